"""A/B of the rows-per-wave of normalise + patchify (CHB_NP_ROWS is read once per process: one subprocess per value)."""
import os
import subprocess
import sys

CODE = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath("%s"))))
from chambers_amd import kernels as K
x = torch.randint(0, 256, (512, 224, 224, 3), dtype=torch.uint8, device="cuda")
out = torch.empty(512 * 196, 768, dtype=torch.bfloat16, device="cuda")
K.normalize_patchify(x, 16, "tf", out=out); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): K.normalize_patchify(x, 16, "tf", out=out)
g.replay(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): g.replay()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 200 * 1e3
print("CHB_NP_ROWS=%%s  %%.1f us  %%.0f GB/s algorithmic" %% (os.environ.get("CHB_NP_ROWS"), us, 231.2e6 / us / 1e3))
''' % os.path.abspath(__file__)
for rows in ("4", "8", "16"):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, CHB_NP_ROWS=rows), capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-500:])
