"""Which clock does the chip hold under the NT GEMMs?  (MI355X_MICROARCH.md "DVFS give-back" 1 and 6.)

Needs the diagnostic build:  tools/ab_build.sh clk gemm.hip -DCHB_CLOCK_STAMPS ;  CHB_AB_LIB=tools/_ab/libchambers_hip_clk.so
python tools/gemm_clock.py [seconds per arm].  Per shape and operand kind (random / zeros): back-to-back launches for the given
time, then the LAST launch's stamps: clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups; the MFMA pipe's
share of that launch = MFMA cycles of a SIMD (2 waves x tiles x K-steps x 64 MFMAs x 8 cycles [16x16x32 bf16: 8 passes of 4... see
DESIGN]) is reported as TFLOP/s at the HELD clock's peak, i.e. achieved / (2.5 PF x clock / 2.4 GHz).
"""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chambers_amd import _build
assert os.environ.get("CHB_AB_LIB"), "run with CHB_AB_LIB=<the -DCHB_CLOCK_STAMPS build>"
_build.LIB_PATH = os.path.abspath(os.environ["CHB_AB_LIB"])
from chambers_amd import _lib, kernels as K

lib = ctypes.CDLL(_build.LIB_PATH)
lib.chb_debug_clock_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
M = 512 * 197
SHAPES = [("qkv_fwd", M, 2304, 768, K.EPI_NONE), ("fc1_fwd", M, 3072, 768, K.EPI_GELU), ("fc2_fwd", M, 768, 3072, K.EPI_RESID),
          ("fc1_dgrad", M, 768, 3072, K.EPI_NONE), ("square_8k", 8192, 8192, 8192, K.EPI_NONE)]


def stamps(n=256):
    buf = np.zeros((n, 4), dtype=np.uint64)
    assert lib.chb_debug_clock_stamps(buf.ctypes.data, n) == 0
    d_clk = (buf[:, 2] - buf[:, 0]).astype(np.float64)
    d_ref = (buf[:, 3] - buf[:, 1]).astype(np.float64)
    ok = d_ref > 0
    return float(np.median(d_clk[ok] / d_ref[ok]) * 100e6), float(np.median(d_ref[ok]) / 100e6)


for name, m, n, k, epi in SHAPES:
    for kind in ("random", "zeros"):
        mk = (lambda *s: torch.randn(*s, device="cuda")) if kind == "random" else (lambda *s: torch.zeros(*s, device="cuda"))
        a, b = mk(m, k).to(torch.bfloat16), mk(n, k).to(torch.bfloat16)
        bias = mk(n)
        out = torch.empty(m, n, dtype=torch.float32 if epi == K.EPI_RESID else torch.bfloat16, device="cuda")
        aux = mk(m, n).to(torch.bfloat16) if epi == K.EPI_GELU else None
        resid = mk(m, n) if epi == K.EPI_RESID else None
        fn = lambda: K.gemm_nt(a, b, out, bias=bias, epilogue=epi, aux=aux, resid=resid, drop_rate=0.1 if epi == K.EPI_RESID else 0.0, drop_key=5)
        fn(); torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < secs:            # hold the load so the clock settles
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        clk, span = stamps()
        tf = 2.0 * m * n * k / ms / 1e9
        peak_at_clk = 2500.0 * clk / 2.4e9
        print("%-10s %-6s  %.3f ms  %7.1f TF/s  clock held %.2f GHz (workgroup span %.3f ms)  = %.2f of the MFMA peak AT THAT CLOCK (%.0f TF/s)"
              % (name, kind, ms, tf, clk / 1e9, span * 1e3, tf / peak_at_clk, peak_at_clk), flush=True)
