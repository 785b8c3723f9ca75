"""Reference ceiling only (NOT used by the product): the vendor GEMM (torch.matmul -> hipBLASLt) and this library's NT kernel on large
square bf16 shapes, N(0,1) operands - how close does a production kernel get to the probe's K-loop ceiling (tools/probe) when the
epilogue and tile quantisation are negligible?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chambers_amd import kernels as K

for n in (4096, 8192, 16384):
    a = torch.randn(n, n, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, n, device="cuda").to(torch.bfloat16)
    out = torch.empty(n, n, dtype=torch.bfloat16, device="cuda")
    for name, f in (("hipBLASLt (torch.matmul, NT)", lambda: torch.matmul(a, b.t())), ("chb_gemm_nt", lambda: K.gemm_nt(a, b, out))):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if n <= 8192 else 5
        s.record()
        for _ in range(reps):
            f()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        print("%5d^3  %-30s %8.3f ms  %7.1f TFLOP/s" % (n, name, ms, 2.0 * n ** 3 / ms / 1e9), flush=True)
