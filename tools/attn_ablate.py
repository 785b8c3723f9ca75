"""Where the lean attention backward spends its time: CHB DEBUG option 1 runs the kernel without its main loop (prologue: 125 KiB of
Q / K / V / dO / O per head into LDS and registers; epilogue: dK / dV stores).  Results are wrong in that mode; only the time is read."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chambers_amd import _lib, kernels as K
B, N, H = 512, 197, 12
D = H * 64
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda")
do = torch.randn(B * N, D, device="cuda").to(torch.bfloat16)
dqkv = torch.empty(B * N, 3 * D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * H * N, device="cuda")
bits = K.attention_drop_bits(B, N, H)


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


K.attention_fwd(qkv, o, lse, B, N, H, 64, 0.1, 7, drop_bits=bits)
for dbg, label in ((0, "whole kernel"), (1, "prologue + epilogue only"), (0, "whole kernel"), (1, "prologue + epilogue only")):
    _lib.set_option("DEBUG", dbg)
    print("attention backward (lean, keep bits), %s: %.3f ms" % (label, t(lambda: K.attention_bwd(qkv, o, do, lse, dqkv, B, N, H, 64, 0.1, 7, drop_bits=bits))), flush=True)
_lib.set_option("DEBUG", 0)
