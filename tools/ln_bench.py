"""LayerNorm forward / backward at the ViT-B/16 B=512 shape (M = 100864 rows x 768), HIP events; CHB_AB_LIB picks an A/B library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chambers_amd import _build
if os.environ.get("CHB_AB_LIB"):
    _build.LIB_PATH = os.path.abspath(os.environ["CHB_AB_LIB"])
from chambers_amd import kernels as K

M, D = 512 * 197, 768
x = torch.randn(M, D, device="cuda")
gamma, beta = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
h = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
dy = torch.randn(M, D, device="cuda").to(torch.bfloat16)
dx = torch.randn(M, D, device="cuda")
dz = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
dg, db, dzs = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


K.layernorm_fwd(x, D, gamma, beta, h, mean, rstd, M, D, 1e-6)
t_f = timed(lambda: K.layernorm_fwd(x, D, gamma, beta, h, mean, rstd, M, D, 1e-6))
t_b = timed(lambda: K.layernorm_bwd(dy, x, D, mean, rstd, gamma, dx, D, True, dg, db, M, D, dz=dz, dz_colsum=dzs, drop_rate=0.1, drop_key=7))
t_b0 = timed(lambda: K.layernorm_bwd(dy, x, D, mean, rstd, gamma, dx, D, False, dg, db, M, D))
fb = M * D * (4 + 2)
bb = M * D * (2 + 4 + 4 + 4 + 2)
print("ln_fwd %.1f us (%.2f TB/s algorithmic)   ln_bwd(accumulate + dz) %.1f us (%.2f TB/s)   ln_bwd(plain) %.1f us (%.2f TB/s)"
      % (t_f, fb / t_f / 1e6, t_b, bb / t_b / 1e6, t_b0, M * D * (2 + 4 + 4) / t_b0 / 1e6))
