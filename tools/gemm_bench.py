"""Micro-benchmark of the GEMM kernels at the ViT-B/16 B=512 shapes (events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chambers_amd import _build
if os.environ.get("CHB_AB_LIB"):        # A/B builds of the library (tools/ab_build.sh), this tool only
    _build.LIB_PATH = os.path.abspath(os.environ["CHB_AB_LIB"])
from chambers_amd import _lib, kernels as K

M = int(os.environ.get("GEMM_BENCH_M", str(512 * 197)))
SHAPES_NT = [("qkv_fwd", M, 2304, 768, K.EPI_NONE), ("proj_fwd", M, 768, 768, K.EPI_RESID), ("fc1_fwd", M, 3072, 768, K.EPI_GELU),
             ("fc2_fwd", M, 768, 3072, K.EPI_RESID), ("fc2_dgrad", M, 3072, 768, K.EPI_DGELU), ("fc1_dgrad", M, 768, 3072, K.EPI_NONE),
             ("qkv_dgrad", M, 768, 2304, K.EPI_NONE)]
SHAPES_TN = [("qkv_wgrad", M, 768, 2304), ("proj_wgrad", M, 768, 768), ("fc1_wgrad", M, 768, 3072), ("fc2_wgrad", M, 3072, 768)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
which = sys.argv[2] if len(sys.argv) > 2 else "all"


def timeit(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


if which in ("all", "nt"):
    for name, m, n, k, epi in SHAPES_NT:
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
        bias = torch.randn(n, device="cuda")
        out = torch.empty(m, n, dtype=torch.float32 if epi == K.EPI_RESID else torch.bfloat16, device="cuda")
        aux = torch.randn(m, n, device="cuda").to(torch.bfloat16) if epi in (K.EPI_GELU, K.EPI_DGELU) else None
        resid = torch.randn(m, n, device="cuda") if epi == K.EPI_RESID else None
        for walk in os.environ.get("GEMM_BENCH_WALKS", "1").split(","):
            _lib.set_option("GEMM_WALK", int(walk))
            ms = timeit(lambda: K.gemm_nt(a, b, out, bias=bias, epilogue=epi, aux=aux, resid=resid, drop_rate=0.1 if epi == K.EPI_RESID else 0.0, drop_key=5))
            print("%-10s M=%d N=%d K=%d epi=%d walk=%s  %.3f ms  %.1f TF/s" % (name, m, n, k, epi, walk, ms, 2.0 * m * n * k / ms / 1e9), flush=True)
if which in ("all", "tn"):
    for name, m, kd, nd in SHAPES_TN:
        x = torch.randn(m, kd, device="cuda").to(torch.bfloat16)
        dy = torch.randn(m, nd, device="cuda").to(torch.bfloat16)
        dw = torch.zeros(kd, nd, device="cuda")
        ms = timeit(lambda: K.gemm_tn(x, dy, dw))
        ws = torch.empty(K.tn_workspace_elems(kd, nd), device="cuda")
        ms2 = timeit(lambda: K.gemm_tn(x, dy, dw, ws=ws))      # partial planes + fold launch (both launches timed)
        print("%-10s M=%d Kd=%d Nd=%d  atomics %.3f ms  %.1f TF/s   planes+fold %.3f ms  %.1f TF/s"
              % (name, m, kd, nd, ms, 2.0 * m * kd * nd / ms / 1e9, ms2, 2.0 * m * kd * nd / ms2 / 1e9), flush=True)
