"""Tile queue of the persistent NT GEMM on / off in ONE process: bit-equal outputs, interleaved timing (tools/gemm_ab.py with the
option GEMM_TILE_QUEUE instead of GEMM_ALGO)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from chambers_amd import _lib, kernels as K

M = 512 * 197
SHAPES = [("qkv_fwd", M, 2304, 768, K.EPI_NONE), ("proj_fwd", M, 768, 768, K.EPI_RESID), ("fc1_fwd", M, 3072, 768, K.EPI_GELU),
          ("fc2_fwd", M, 768, 3072, K.EPI_RESID), ("fc2_dgrad", M, 3072, 768, K.EPI_DGELU), ("fc1_dgrad", M, 768, 3072, K.EPI_NONE),
          ("ragged", 6304 + 64, 768, 768, K.EPI_NONE), ("k640", 8192, 1024, 640, K.EPI_NONE), ("k512", 8192, 1024, 512, K.EPI_NONE), ("k448 (static)", 8192, 1024, 448, K.EPI_NONE)]
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def timed(q, fn, iters=10):
    _lib.set_option("GEMM_TILE_QUEUE", q)
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


bad = 0
for name, m, n, k, epi in SHAPES:
    torch.manual_seed(2)
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    bias = torch.randn(n, device="cuda")
    odt = torch.float32 if epi == K.EPI_RESID else torch.bfloat16
    aux = torch.randn(m, n, device="cuda").to(torch.bfloat16) if epi in (K.EPI_GELU, K.EPI_DGELU) else None
    resid = torch.randn(m, n, device="cuda") if epi == K.EPI_RESID else None
    kw = dict(bias=bias, epilogue=epi, aux=aux, resid=resid, drop_rate=0.1 if epi == K.EPI_RESID else 0.0, drop_key=5)
    outs = []
    for q, reps in ((0, 1), (1, 40)):          # 40 launches with the queue: 40 different slots, every counter must come back to zero
        _lib.set_option("GEMM_TILE_QUEUE", q)
        for _ in range(reps):
            out = torch.full((m, n), float("nan"), dtype=odt, device="cuda")
            K.gemm_nt(a, b, out, **kw)
            torch.cuda.synchronize()
            outs.append(out)
    same = all(torch.equal(o, outs[0]) for o in outs[1:])
    bad += 0 if same else 1
    f = lambda: K.gemm_nt(a, b, outs[0], **kw)      # noqa: E731
    t0, t1 = [], []
    for _ in range(ROUNDS):
        t0.append(timed(0, f))
        t1.append(timed(1, f))
    fl = 2.0 * m * n * k / 1e9
    print("%-14s M=%d N=%d K=%d epi=%d  static %.3f ms (%.0f TF/s) | queue %.3f ms (%.0f TF/s) | %+.1f%%  %s"
          % (name, m, n, k, epi, np.median(t0), fl / np.median(t0), np.median(t1), fl / np.median(t1), 100 * (np.median(t0) / np.median(t1) - 1),
             "bit-equal x40" if same else "DIFFERS"), flush=True)
_lib.set_option("GEMM_TILE_QUEUE", 0)
sys.exit(1 if bad else 0)
