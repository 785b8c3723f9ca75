"""Persistent pipelined attention forward (attn_fwd_pipe_kernel) against the whole-head kernel (CHB_ATTN_FWD_ALGO = 3): bit equality of
o, lse and the keep bits, then timings at the bench shape.    python tools/attn_fwd_pipe_check.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chambers_amd import _lib, kernels as K


def run(B, N, H, rate, algo, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    D = H * 64
    qkv = torch.randn(B * N, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
    o = torch.full((B * N, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.full((B * H * N,), float("nan"), device="cuda")
    bits = K.attention_drop_bits(B, N, H) if rate else None
    if bits is not None:
        bits.fill_(-1)
    _lib.set_option("ATTN_FWD_ALGO", algo)
    try:
        K.attention_fwd(qkv, o, lse, B, N, H, 64, rate, 7, drop_bits=bits)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("ATTN_FWD_ALGO", 0)
    return o, lse, bits


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


for (B, N, H, rate) in [(2, 197, 3, 0.0), (2, 197, 3, 0.1), (1, 193, 1, 0.1), (3, 198, 2, 0.1), (2, 208, 2, 0.25), (23, 197, 12, 0.1), (23, 197, 12, 0.0),
                        (64, 197, 12, 0.1), (100, 200, 7, 0.1)]:
    a = run(B, N, H, rate, 3)
    b = run(B, N, H, rate, 0)
    same = torch.equal(a[0].view(torch.int16), b[0].view(torch.int16)) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)) and \
        (a[2] is None or torch.equal(a[2], b[2]))
    print("B %3d N %3d H %2d rate %.2f: pipe == whole-head bitwise (o, lse, bits): %s" % (B, N, H, rate, same), flush=True)
    if not same:
        print("  o", torch.equal(a[0].view(torch.int16), b[0].view(torch.int16)), "lse", torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)),
              "nan in pipe o", bool(torch.isnan(b[0].float()).any()))
        sys.exit(1)

B, N, H = 512, 197, 12
D = H * 64
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * H * N, device="cuda")
for rate in (0.0, 0.1):
    bits = K.attention_drop_bits(B, N, H) if rate else None
    res = {}
    for algo, name in ((3, "whole-head"), (0, "pipe")):
        _lib.set_option("ATTN_FWD_ALGO", algo)
        res[name] = t(lambda: K.attention_fwd(qkv, o, lse, B, N, H, 64, rate, 7, drop_bits=bits))
    _lib.set_option("ATTN_FWD_ALGO", 0)
    hbm = (B * N * 4 * D * 2) / 1e9
    print("B 512 N 197 H 12 rate %.1f: whole-head %.3f ms (%.2f TB/s)   pipe %.3f ms (%.2f TB/s)" % (rate, res["whole-head"], hbm / res["whole-head"], res["pipe"], hbm / res["pipe"]), flush=True)
