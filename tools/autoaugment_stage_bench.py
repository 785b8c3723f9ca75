"""The augmentation STAGE under AutoAugment (policy v0, batch-shared): the 25 sub-policies x the four chance outcomes, fused into the
normalise + patchify pass (chb_aug_fused, patch = 16), on BASELINE config 5's batch [128,384,384,3] by default; the expectation over
the scheme's own draws (sub-policy uniform, each step applied with its probability) beside the plain mean.  HIP-graph replay timing.
usage: python tools/autoaugment_stage_bench.py [B] [size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from chambers_amd import augmentations as aug
from chambers_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 384
REP = 5
x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
patches = torch.empty((B * (H // 16) * (W // 16), 768), dtype=torch.bfloat16, device="cuda")
stage_bytes = 3.0 * B * H * W * 3
layer = aug.AutoAugment()
policy = aug.augmentation_schemes._AUTO_AUGMENT_POLICY_V0


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(REP):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    graph.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REP * 1e3


t_patch = timed(lambda: K.normalize_patchify(x, 16, "tf", out=patches))
rows, expect = [], 0.0
for pol in range(25):
    sub = policy[pol]
    t = {}
    for apply in ((True, True), (True, False), (False, True), (False, False)):
        plan = layer.plan(x.shape, {"policy": pol, "apply": apply, "negate": (False, True)})
        t[apply] = timed(lambda: K.aug_fused(x, plan, patch=16, out=patches)) if len(plan) else t_patch
    p0, p1 = float(sub[0][1]), float(sub[1][1])
    e = p0 * p1 * t[(True, True)] + p0 * (1 - p1) * t[(True, False)] + (1 - p0) * p1 * t[(False, True)] + (1 - p0) * (1 - p1) * t[(False, False)]
    expect += e / 25.0
    rows.append((pol, "%s(%.1f) > %s(%.1f)" % (sub[0][0], p0, sub[1][0], p1), t[(True, True)], e))
gb = lambda us: stage_bytes / us / 1e3      # noqa: E731
print("stage = AutoAugment v0 sub-policy -> normalise('tf') -> bf16 patch rows, batch [%d,%d,%d,3]; algorithmic bytes %.1f MB" % (B, H, W, stage_bytes / 1e6))
print("normalise + patchify alone            %8.1f us  %7.1f GB/s  %5.1f %% of 8 TB/s" % (t_patch, gb(t_patch), 100 * gb(t_patch) / 8000))
both = np.mean([r[2] for r in rows])
print("fused, both steps applied, mean of 25 %8.1f us  %7.1f GB/s  %5.1f %%" % (both, gb(both), 100 * gb(both) / 8000))
print("fused, expectation over the draws     %8.1f us  %7.1f GB/s  %5.1f %%" % (expect, gb(expect), 100 * gb(expect) / 8000))
for pol, name, tt, e in rows:
    print("  %2d  %-44s both applied %7.1f us   expected %7.1f us" % (pol, name, tt, e))
