"""Run a few op pairs of the fused augmentation stage a fixed number of times (for rocprofv3 --kernel-trace --stats: which kernels a
pair launches and how long each takes).  usage: python tools/aug_pair_probe.py First>Second [First>Second ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from chambers_amd import augmentations as aug
from chambers_amd import kernels as K

B, H, W = 512, 224, 224
x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
patches = torch.empty((B * (H // 16) * (W // 16), 768), dtype=torch.bfloat16, device="cuda")
g = np.random.Generator(np.random.PCG64(0))
centers = torch.as_tensor(np.stack([g.integers(0, H, size=B), g.integers(0, W, size=B)], axis=1).astype(np.int32), device="cuda")
layer = aug.RandAugment(2, 9)
names = layer._OPS
for pair in sys.argv[1:]:
    a, b = pair.split(">")
    dec = [{"op": names.index(a), "negate": False, "centers": centers}, {"op": names.index(b), "negate": True, "centers": centers}]
    plan = layer.plan(x.shape, dec)
    for _ in range(20):
        K.aug_fused(x, plan, patch=16, out=patches)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        K.aug_fused(x, plan, patch=16, out=patches)
    e.record()
    torch.cuda.synchronize()
    print("%-28s %.1f us per stage (eager launches)" % (pair, s.elapsed_time(e) / 20 * 1e3))
