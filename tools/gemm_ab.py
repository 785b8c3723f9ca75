"""A/B of two NT GEMM schedules in ONE process: outputs compared bit for bit (same K order, same accumulation order), then
interleaved timing rounds (median / min).   python tools/gemm_ab.py [algo_a algo_b] [rounds]
algo ids = CHB_OPT_GEMM_ALGO: 2 = persistent 256x256 (lockstep waves), 4 = persistent 256x256 ping-pong."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from chambers_amd import _build
if os.environ.get("CHB_AB_LIB"):        # A/B builds of the library (tools/ab_build.sh), this tool only
    _build.LIB_PATH = os.path.abspath(os.environ["CHB_AB_LIB"])
from chambers_amd import _lib, kernels as K

A_ALGO = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B_ALGO = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 7
BATCH = int(os.environ.get("AB_BATCH", "512"))
M = BATCH * 197
SHAPES = [("qkv_fwd", M, 2304, 768, K.EPI_NONE), ("proj_fwd", M, 768, 768, K.EPI_RESID), ("fc1_fwd", M, 3072, 768, K.EPI_GELU),
          ("fc2_fwd", M, 768, 3072, K.EPI_RESID), ("fc2_dgrad", M, 3072, 768, K.EPI_DGELU), ("fc1_dgrad", M, 768, 3072, K.EPI_NONE),
          ("qkv_dgrad", M, 768, 2304, K.EPI_NONE), ("proj_dgrad", M, 768, 768, K.EPI_NONE)]


def run(algo, a, b, out, **kw):
    _lib.set_option("GEMM_ALGO", algo)
    K.gemm_nt(a, b, out, **kw)


def timed(algo, a, b, out, iters, **kw):
    _lib.set_option("GEMM_ALGO", algo)
    K.gemm_nt(a, b, out, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        K.gemm_nt(a, b, out, **kw)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


bad = 0
for name, m, n, k, epi in SHAPES:
    torch.manual_seed(1)
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    bias = torch.randn(n, device="cuda")
    odt = torch.float32 if epi == K.EPI_RESID else torch.bfloat16
    aux_in = torch.randn(m, n, device="cuda").to(torch.bfloat16) if epi == K.EPI_DGELU else None
    resid = torch.randn(m, n, device="cuda") if epi == K.EPI_RESID else None
    colsum = epi == K.EPI_DGELU or name == "proj_dgrad"
    outs = {}
    for algo in (A_ALGO, B_ALGO):
        res = []
        for rep in range(3 if algo == B_ALGO else 1):          # the new schedule several times: a race shows as a run that differs
            out = torch.full((m, n), float("nan"), dtype=odt, device="cuda")
            aux = aux_in.clone() if aux_in is not None else (torch.empty(m, n, dtype=torch.bfloat16, device="cuda") if epi == K.EPI_GELU else None)
            cs = torch.zeros(n, device="cuda") if colsum else None
            run(algo, a, b, out, bias=bias, epilogue=epi, aux=aux, resid=resid, drop_rate=0.1 if epi == K.EPI_RESID else 0.0, drop_key=5, colsum=cs)
            torch.cuda.synchronize()
            res.append((out, aux if epi == K.EPI_GELU else None, cs))
        outs[algo] = res
    ref = outs[A_ALGO][0]
    ok = True
    for rep, got in enumerate(outs[B_ALGO]):
        same = torch.equal(got[0], ref[0]) and (ref[1] is None or torch.equal(got[1], ref[1]))
        cs_ok = ref[2] is None or torch.allclose(got[2], ref[2], rtol=1e-4, atol=1e-2)
        if not (same and cs_ok):
            ok = False
            d = (got[0].float() - ref[0].float())
            nbad = int((d != 0).sum() + torch.isnan(d).sum())
            print("   MISMATCH %s rep %d: %d elements differ (max %.3g), colsum ok %s" % (name, rep, nbad, float(d.nan_to_num(1e30).abs().max()), cs_ok))
    bad += 0 if ok else 1
    ta, tb = [], []
    aux = aux_in if aux_in is not None else (torch.empty(m, n, dtype=torch.bfloat16, device="cuda") if epi == K.EPI_GELU else None)
    out = torch.empty(m, n, dtype=odt, device="cuda")
    for _ in range(ROUNDS):
        ta.append(timed(A_ALGO, a, b, out, 10, bias=bias, epilogue=epi, aux=aux, resid=resid, drop_rate=0.1 if epi == K.EPI_RESID else 0.0, drop_key=5))
        tb.append(timed(B_ALGO, a, b, out, 10, bias=bias, epilogue=epi, aux=aux, resid=resid, drop_rate=0.1 if epi == K.EPI_RESID else 0.0, drop_key=5))
    fl = 2.0 * m * n * k / 1e9
    print("%-10s M=%d N=%d K=%d epi=%d  algo %d: median %.3f ms (%.0f TF/s) min %.3f | algo %d: median %.3f ms (%.0f TF/s) min %.3f | %+.1f%%  %s"
          % (name, m, n, k, epi, A_ALGO, np.median(ta), fl / np.median(ta), min(ta), B_ALGO, np.median(tb), fl / np.median(tb), min(tb),
             100.0 * (np.median(ta) / np.median(tb) - 1.0), "bit-equal" if ok else "DIFFERS"), flush=True)
_lib.set_option("GEMM_ALGO", 0)
sys.exit(1 if bad else 0)
