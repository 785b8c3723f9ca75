"""LayerNorm forward / backward timings at the step's shape: python tools/ln_bench.py [M D].  CHB_AB_LIB picks an A/B build."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chambers_amd import _build
if os.environ.get("CHB_AB_LIB"):
    _build.LIB_PATH = os.path.abspath(os.environ["CHB_AB_LIB"])
from chambers_amd import kernels as K

M, D = (int(a) for a in sys.argv[1:3]) if len(sys.argv) >= 3 else (512 * 197, 768)
x = torch.randn(M, D, device="cuda")
gamma, beta = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
y = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
dy = torch.randn(M, D, device="cuda").to(torch.bfloat16)
dx = torch.randn(M, D, device="cuda")
dz = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
dg, db, dzs = (torch.zeros(D, device="cuda") for _ in range(3))
# a GEMM-sized filler between timed launches so that nothing is served from the Infinity Cache
fill_a = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device="cuda")


def t(fn, it=20):
    tot = 0.0
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    for _ in range(it):
        fill_a.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / it * 1e3


from chambers_amd import _lib
_lib.set_option("LN_STREAM", int(os.environ.get("LN_STREAM", "0")))
f = t(lambda: K.layernorm_fwd(x, D, gamma, beta, y, mean, rstd, M, D, 1e-6))
b = t(lambda: K.layernorm_bwd(dy, x, D, mean, rstd, gamma, dx, D, True, dg, db, M, D, dz=dz, dz_colsum=dzs, drop_rate=0.1, drop_key=7))
fb, bb = M * D * 6 / 1e6, M * D * (2 + 4 + 4 + 4 + 2) / 1e6
print("LN_STREAM=%s %s  fwd %.1f us (%.2f TB/s)   bwd %.1f us (%.2f TB/s)" % (os.environ.get("LN_STREAM", "0"), os.environ.get("CHB_AB_LIB", "default"), f, fb / f, b, bb / b))
