"""Where do the cycles of gemm_nt256_kernel's K-step go?  Diagnostic build only:
tools/ab_build.sh ph gemm.hip -DCHB_PHASE_STAMPS -DCHB_CLOCK_STAMPS ; CHB_AB_LIB=tools/_ab/libchambers_hip_ph.so python tools/gemm_phases.py

Every wave sums the s_memtime cycles of seven segments of the K-step; waves 0 (rows 0-127, SIMD 0) and 5 (rows 128-255, SIMD 1)
of each workgroup report.  Printed: median over workgroups of cycles per K-step and segment, beside the 2 x 32 x 16 = 1024 cycles
per phase the two waves of a SIMD need on the MFMA pipe.  The stamps cost about 10 % of the loop: read the SHARES, not the total.
"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chambers_amd import _build
assert os.environ.get("CHB_AB_LIB")
_build.LIB_PATH = os.path.abspath(os.environ["CHB_AB_LIB"])
from chambers_amd import _lib, kernels as K

lib = ctypes.CDLL(_build.LIB_PATH)
lib.chb_debug_phase_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
M = 512 * 197
SHAPES = [("qkv_fwd", M, 2304, 768, K.EPI_NONE), ("fc1_dgrad", M, 768, 3072, K.EPI_NONE), ("fc2_fwd", M, 768, 3072, K.EPI_RESID),
          ("square_8k", 8192, 8192, 8192, K.EPI_NONE)]
NAMES = ["bookkeeping", "stageA+16 reads+wait", "32 MFMA issue", "8 reads+wait", "vmcnt(0)", "barrier", "stageB+32 MFMA issue",
         "pre-epilogue", "epilogue"]
if os.environ.get("PHASES_ALGO", "2") == "5":      # the pipelined kernel stamps fewer points (cycles summed over the tile, shown per K-step)
    NAMES = ["-", "Q3(prev)+Q0+Q1+Q2 incl. barrier 1", "step-end vmcnt, K-step 1 of a tile", "step-end vmcnt, K-step 0", "step-end vmcnt, K-steps 3..",
             "barrier 2", "step-end vmcnt, K-step 2", "pre-epilogue", "epilogue"]
for name, m, n, k, epi in SHAPES:
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    bias = torch.randn(n, device="cuda")
    out = torch.empty(m, n, dtype=torch.float32 if epi == K.EPI_RESID else torch.bfloat16, device="cuda")
    resid = torch.randn(m, n, device="cuda") if epi == K.EPI_RESID else None
    _lib.set_option("GEMM_ALGO", int(os.environ.get("PHASES_ALGO", "2")))
    fn = lambda: K.gemm_nt(a, b, out, bias=bias, epilogue=epi, resid=resid, drop_rate=0.1 if epi == K.EPI_RESID else 0.0, drop_key=5)
    for _ in range(200):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    buf = np.zeros((256, 2, 10), dtype=np.uint32)
    assert lib.chb_debug_phase_stamps(buf.ctypes.data, 256) == 0
    print("%s M=%d N=%d K=%d: %.3f ms (%.0f TF/s with the stamps in)" % (name, m, n, k, ms, 2.0 * m * n * k / ms / 1e9))
    ksteps = k // 64
    for wv, label in ((0, "wave 0"), (1, "wave 5")):
        d = buf[:, wv, :].astype(np.float64)
        tiles = d[:, 9]
        ok = tiles > 0
        per_step = d[ok, :7] / (tiles[ok, None] * ksteps)
        per_tile = d[ok, 7:9] / tiles[ok, None]
        med = np.median(per_step, axis=0)
        print("  %s: tiles/workgroup %.1f  K-step total %.0f cycles (MFMA floor 2048)" % (label, np.median(tiles[ok]), med.sum()))
        for nm, v in zip(NAMES[:7], med):
            print("      %-24s %6.0f" % (nm, v))
        print("      per tile: %s %.0f, %s %.0f cycles (= %.1f K-steps' worth)" % (NAMES[7], np.median(per_tile[:, 0]), NAMES[8], np.median(per_tile[:, 1]),
                                                                                   np.median(per_tile[:, 1]) / max(med.sum(), 1.0)))
    sys.stdout.flush()
