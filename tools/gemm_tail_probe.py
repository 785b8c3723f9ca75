"""Does it pay to run the partial last round of the persistent NT GEMM as smaller tiles?  python tools/gemm_tail_probe.py [M]
For each shape: one launch over all M rows (default kernel) against main rows (whole rounds of 256x256 tiles on 256 CUs) + the
remaining row tiles through the 128x256 two-workgroups-per-CU kernel (GEMM_ALGO 3) or the 128x128 kernel (GEMM_ALGO 1)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chambers_amd import _lib, kernels as K

M = int(sys.argv[1]) if len(sys.argv) > 1 else 101120
CUS = torch.cuda.get_device_properties(0).multi_processor_count


def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for n, k in ((1024, 1024), (1024, 3072), (1024, 4096), (768, 768), (768, 2304), (768, 3072)):
    a = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    b = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
    out = torch.empty(M, n, dtype=torch.bfloat16, device="cuda")
    tiles_m, tiles_n = M // 256, n // 256
    rounds = (tiles_m * tiles_n) // CUS
    main_rows = (rounds * CUS // tiles_n) * 256
    tail_rows = M - main_rows
    _lib.set_option("GEMM_ALGO", 0)
    whole = t(lambda: K.gemm_nt(a, b, out))
    main = t(lambda: K.gemm_nt(a[:main_rows], b, out[:main_rows]))
    line = "N %4d K %4d: %d tiles = %.2f rounds | whole %.1f us | main (%d rows) %.1f" % (n, k, tiles_m * tiles_n, tiles_m * tiles_n / CUS, whole, main_rows, main)
    for algo in (3, 1, 0):
        _lib.set_option("GEMM_ALGO", algo)
        tail = t(lambda: K.gemm_nt(a[main_rows:], b, out[main_rows:]))

        def both():
            _lib.set_option("GEMM_ALGO", 0)
            K.gemm_nt(a[:main_rows], b, out[:main_rows])
            _lib.set_option("GEMM_ALGO", algo)
            K.gemm_nt(a[main_rows:], b, out[main_rows:])
        line += " | tail (%d rows) algo %d: %.1f, both %.1f" % (tail_rows, algo, tail, t(both))
    _lib.set_option("GEMM_ALGO", 0)
    print(line, flush=True)
