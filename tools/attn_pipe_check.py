"""Persistent pipelined attention backward (attn_bwd_pipe_kernel) against the lean one-workgroup-per-head kernel: bit equality of
dQ / dK / dV on several shapes (the two run the same arithmetic in the same order), then timings at the bench shape.
    python tools/attn_pipe_check.py [--time-only]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chambers_amd import _build
if os.environ.get("CHB_AB_LIB"):        # A/B builds of the library (tools/ab_build.sh), this tool only
    _build.LIB_PATH = os.path.abspath(os.environ["CHB_AB_LIB"])
    _build.is_current = lambda: True
from chambers_amd import _lib, kernels as K


def run(B, N, H, rate, algo, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    D = H * 64
    qkv = torch.randn(B * N, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(B * N, D, device="cuda", generator=g).to(torch.bfloat16)
    o = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B * H * N, device="cuda")
    bits = K.attention_drop_bits(B, N, H) if rate else None
    dqkv = torch.full((B * N, 3 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
    K.attention_fwd(qkv, o, lse, B, N, H, 64, rate, 7, drop_bits=bits)
    _lib.set_option("ATTN_BWD_ALGO", algo)
    try:
        K.attention_bwd(qkv, o, do, lse, dqkv, B, N, H, 64, rate, 7, drop_bits=bits)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("ATTN_BWD_ALGO", 0)
    return dqkv


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


if "--time-only" not in sys.argv:
    for (B, N, H, rate) in [(2, 197, 3, 0.0), (2, 197, 3, 0.1), (1, 193, 1, 0.1), (3, 198, 2, 0.1), (2, 208, 2, 0.25), (23, 197, 12, 0.1),
                            (23, 197, 12, 0.0), (64, 197, 12, 0.1), (100, 200, 7, 0.1)]:
        a = run(B, N, H, rate, 4)
        b = run(B, N, H, rate, 0)
        same = torch.equal(a.view(torch.int16), b.view(torch.int16))
        print("B %3d N %3d H %2d rate %.2f: pipe == lean bitwise: %s%s" % (B, N, H, rate, same, "" if same else "  max|diff| %g, nan in pipe %s" % (
            float((a.float() - b.float()).abs().nan_to_num(1e9).max()), bool(torch.isnan(b.float()).any()))), flush=True)
        if not same:
            sys.exit(1)

B, N, H = 512, 197, 12
D = H * 64
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
do = torch.randn(B * N, D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * H * N, device="cuda")
dqkv = torch.empty(B * N, 3 * D, dtype=torch.bfloat16, device="cuda")
for rate in (0.0, 0.1):
    bits = K.attention_drop_bits(B, N, H) if rate else None
    K.attention_fwd(qkv, o, lse, B, N, H, 64, rate, 7, drop_bits=bits)
    res = {}
    for algo, name in ((4, "lean"), (0, "pipe")):
        _lib.set_option("ATTN_BWD_ALGO", algo)
        res[name] = t(lambda: K.attention_bwd(qkv, o, do, lse, dqkv, B, N, H, 64, rate, 7, drop_bits=bits))
    _lib.set_option("ATTN_BWD_ALGO", 0)
    if "--barrier-experiment" in sys.argv:     # timing only, results wrong: the pipe kernel with a barrier every k-th step
        for k, what in ((1000, "no barriers"), (1024, "no phase B"), (2048, "no producer"), (4096, "no dK/dV accumulation"), (8192, "no softmax VALU"),
                        (4096 + 8192, "no acc, no softmax"), (16384, "producer issues no Q pieces (10 instead of 14)"), (16384 + 4096 + 8192 + 1024, "that, no acc/softmax/B"), (4096 + 8192 + 1024, "producer + A's S/dP only"), (1024 + 2048, "A only"), (1024 + 2048 + 4096 + 8192, "A: S/dP MFMAs + dS store only"),
                        (1024 + 2048 + 4096 + 8192 + 1000, "the same without barriers")):
            _lib.set_option("DEBUG", k)
            res["pipe, " + what] = t(lambda: K.attention_bwd(qkv, o, do, lse, dqkv, B, N, H, 64, rate, 7, drop_bits=bits))
        _lib.set_option("DEBUG", 0)
        print({k: round(v, 3) for k, v in res.items()}, flush=True)
    hbm = (B * N * (3 * D + 2 * D + 3 * D) * 2) / 1e9
    print("B 512 N 197 H 12 rate %.1f: lean %.3f ms (%.2f TB/s)   pipe %.3f ms (%.2f TB/s)" % (rate, res["lean"], hbm / res["lean"], res["pipe"], hbm / res["pipe"]), flush=True)
