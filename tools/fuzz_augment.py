"""Randomised parity sweep of the fused augmentation stage against the oracle (beyond the fixed cases of tests/): random shapes (ragged
widths, tiny and wide images), chains of 1-4 RandAugment ops with random signs and cutout centres, batch-shared (cut / uncut / patch rows)
and per-image chains.  usage: python tools/fuzz_augment.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import augment_ref as A
from chambers_amd import augmentations as aug
from chambers_amd import kernels as K

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
bad = 0


def patch_rows(x_u8, p):
    b, h, w, _ = x_u8.shape
    gh, gw = h // p, w // p
    f = A.imagenet_normalize(x_u8[:, :gh * p, :gw * p], "tf").reshape(b, gh, p, gw, p, 3).transpose(0, 1, 3, 2, 4, 5).reshape(b * gh * gw, p * p * 3)
    return torch.from_numpy(np.ascontiguousarray(f)).to(torch.bfloat16)


for case in range(cases):
    b = int(g.integers(1, 5))
    h = int(g.choice([1, 2, 3, 5, 8, 13, 16, 21, 32, 40, 57]))
    w = int(g.choice([1, 2, 4, 5, 7, 12, 16, 23, 32, 48, 50, 64, 70, 130, 252, 260]))
    n = int(g.integers(1, 5))
    shape = (b, h, w, 3)
    x = g.integers(0, 256, size=shape, dtype=np.uint8)
    kind = int(g.integers(0, 4))
    if kind == 1:
        x[0] = (x[0] // 7) + 90          # low contrast
    elif kind == 2:
        x[0] = int(g.integers(0, 256))   # constant image
    elif kind == 3:
        x[..., : max(1, w // 3), :] = 255
    ops = [int(v) for v in g.integers(0, 16, size=n)]
    dec = [{"op": op, "negate": bool(g.uniform() < 0.5),
            "centers": np.stack([g.integers(0, h, size=b), g.integers(0, w, size=b)], axis=1).astype(np.int32)} for op in ops]
    layer = aug.RandAugment(n, 9)
    ref = A.rand_augment(x, n, 9, dec)
    plan = layer.plan(shape, dec)
    xd = torch.as_tensor(x, device="cuda")
    what = []
    if not np.array_equal(K.aug_fused(xd, plan, scratch=True).cpu().numpy(), ref):
        what.append("cut")
    if not np.array_equal(K.aug_fused(xd, plan, scratch=False).cpu().numpy(), ref):
        what.append("uncut")
    p = 4 if (h >= 4 and w >= 4) else 0
    if p and not torch.equal(K.aug_fused(xd, plan, patch=p).cpu().view(torch.int16), patch_rows(ref, p).view(torch.int16)):
        what.append("patch rows")
    # per-image chains: every image its own ops
    pdec = [[{"op": int(g.integers(0, 16)), "negate": bool(g.uniform() < 0.5), "centers": (int(g.integers(0, h)), int(g.integers(0, w)))} for _ in range(n)]
            for _ in range(b)]
    el = aug.RandAugment(n, 9, elementwise=True)
    eref = A.rand_augment_elementwise(x, n, 9, [[dict(d, centers=np.array([d["centers"]], dtype=np.int32)) for d in ds] for ds in pdec])
    ip = el.items_plan(shape, pdec)
    if not np.array_equal(K.aug_fused_items(xd, ip).cpu().numpy(), eref):
        what.append("per-image")
    if p and not torch.equal(K.aug_fused_items(xd, ip, patch=p).cpu().view(torch.int16), patch_rows(eref, p).view(torch.int16)):
        what.append("per-image patch rows")
    if what:
        bad += 1
        print("MISMATCH case %d shape %s ops %s negate %s: %s | per-image ops %s" % (case, shape, ops, [d["negate"] for d in dec], what,
                                                                                  [[d["op"] for d in ds] for ds in pdec]), flush=True)
print("%d cases, %d mismatching" % (cases, bad))
sys.exit(1 if bad else 0)
