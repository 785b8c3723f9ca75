"""The augmentation STAGE of the train step - RandAugment(2, 9) chain -> ImageNetNormalization("tf") -> bf16 patch rows - timed three
ways on a [B,224,224,3] uint8 batch, over all 16 x 16 ordered op pairs (the scheme draws both slots uniformly):
  fused        chb_aug_fused(patch=16): the chain evaluated inside the patchify pass (+ a histogram pass per table op)
  op-by-op     one launch per op, then chb_normalize_patchify_bf16
  elementwise  per-image decisions: every image's chain in one launch (chb_aug_fused_items); and the r02 route: chb_aug_dispatch per
               slot, then chb_normalize_patchify_bf16
Algorithmic bytes of the stage = uint8 batch read once + bf16 patch rows written once = 3 * B*H*W*3, whatever the chain.
Every configuration is replayed from a HIP graph (the kernels are shorter than a Python layer call).
  CHB_STAGE_FUSED_ONLY=1: time the fused path alone and print its table (A/B builds through CHB_AB_LIB: tools/ab_build.sh)."""
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

if os.environ.get("CHB_AB_LIB"):
    from chambers_amd import _build
    _build.LIB_PATH = os.path.abspath(os.environ["CHB_AB_LIB"])
FUSED_ONLY = os.environ.get("CHB_STAGE_FUSED_ONLY", "0") == "1"
ELEM_ONLY = os.environ.get("CHB_STAGE_ELEMENTWISE_ONLY", "0") == "1"      # the per-image stage alone (for rocprofv3 --kernel-trace)
from chambers_amd import augmentations as aug
from chambers_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 224
REP = 5
x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
patches = torch.empty((B * (H // 16) * (W // 16), 768), dtype=torch.bfloat16, device="cuda")
stage_bytes = 3.0 * B * H * W * 3
g = np.random.Generator(np.random.PCG64(0))
centers = torch.as_tensor(np.stack([g.integers(0, H, size=B), g.integers(0, W, size=B)], axis=1).astype(np.int32), device="cuda")
layer = aug.RandAugment(2, 9)
names = layer._OPS


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
    return s.elapsed_time(e) / REP * 1e3     # us


def decisions(a, b):
    return [{"op": a, "negate": False, "centers": centers}, {"op": b, "negate": True, "centers": centers}]


rows = {}
for a, b in itertools.product(range(16), range(16)):
    if ELEM_ONLY:
        break
    dec = decisions(a, b)
    plan = layer.plan(x.shape, dec)
    t_fused = timed(lambda: K.aug_fused(x, plan, patch=16, out=patches))
    if FUSED_ONLY:
        rows["%s>%s" % (names[a], names[b])] = (t_fused, t_fused)
        continue
    layer._transform.fused = False
    t_ops = timed(lambda: K.normalize_patchify(layer(x, training=True, decisions=dec), 16, "tf", out=patches))
    layer._transform.fused = True
    rows["%s>%s" % (names[a], names[b])] = (t_fused, t_ops)

if FUSED_ONLY:
    fused = np.array([v[0] for v in rows.values()])
    print("fused, mean of 256 pairs %.1f us; min %.1f (%s) max %.1f (%s)" % (fused.mean(), fused.min(), list(rows)[int(fused.argmin())], fused.max(),
                                                                                 list(rows)[int(fused.argmax())]))
    print("%-13s" % "" + "".join("%7s" % n[:6] for n in names))
    for a in names:
        print("%-13s" % a + "".join("%7.0f" % rows[a + ">" + b][0] for b in names))
    sys.exit(0)

# elementwise: every image its own pair (uniform draws), resident op records
items = np.zeros((2, B), dtype=K.AUG_ITEM_DTYPE)
for n in range(B):
    for s_ in range(2):
        t = layer.transforms[int(g.integers(0, 16))]
        items[s_, n] = t.dispatch_item(H, W, negate=bool(g.uniform() < 0.5), centers=(int(g.integers(0, H)), int(g.integers(0, W))))
items_dev = [torch.as_tensor(items[s_].view(np.uint8).reshape(B, 64), device="cuda") for s_ in range(2)]
from chambers_amd import _lib          # noqa: E402
n_stats = [int(np.isin(items[s_]["op"], (_lib.AUG_AUTOCONTRAST, _lib.AUG_EQUALIZE)).sum()) for s_ in range(2)]
ws = torch.empty(B * 768, dtype=torch.int32, device="cuda")
bufs = [torch.empty_like(x), torch.empty_like(x)]


def elementwise_stage_r02():          # rounds 1-2: one dispatch launch per slot (+ statistics), then the patchify pass
    src = x
    for s_ in range(2):
        _lib.call("chb_aug_dispatch", _lib.ptr(src), _lib.ptr(bufs[s_]), B, H, W, _lib.ptr(items_dev[s_]), n_stats[s_], _lib.ptr(ws), K._s())
        src = bufs[s_]
    K.normalize_patchify(src, 16, "tf", out=patches)


items_plan = K.AugItemsPlan(items)
items_plan.resident(x.device, H, W)
torch.cuda.synchronize()


def elementwise_stage():              # round 3: every image's chain inside the patchify pass, images sorted by what the chain needs
    K.aug_fused_items(x, items_plan, patch=16, out=patches)


def elementwise_stage_unsorted():     # the same with every image through the general evaluators (chb_aug_fused_items, one final launch)
    import ctypes
    dev_items, cen, tables, _o, _c = items_plan.resident(x.device, H, W)
    cptr = (ctypes.c_void_p * 2)()
    for l, c_ in enumerate(cen):
        if c_ is not None:
            cptr[l] = c_.data_ptr()
    _lib.call("chb_aug_fused_items", _lib.ptr(x), _lib.ptr(patches), B, H, W, 2, _lib.ptr(dev_items), ctypes.cast(cptr, ctypes.c_void_p), tables,
              _lib.ptr(ws_items), 16, K._s())


ws_items = torch.empty(_lib.aug_fused_workspace_ints(B, H, W, 2), dtype=torch.int32, device="cuda")
t_elem_r02 = timed(elementwise_stage_r02)
t_elem = timed(elementwise_stage)
t_elem_unsorted = timed(elementwise_stage_unsorted)
if ELEM_ONLY and os.environ.get("CHB_STAGE_TRACE", "0") == "1":      # one replay of each route, for a kernel trace
    sys.exit(0)
if ELEM_ONLY:
    print("elementwise, per-image chains: sorted by group %.1f us; one launch %.1f us; dispatch x2 %.1f us; groups %s"
          % (t_elem, t_elem_unsorted, t_elem_r02, items_plan.resident(x.device, H, W)[4].reshape(-1, K.ITEMS_GROUPS).tolist()))
    sys.exit(0)
t_patch = timed(lambda: K.normalize_patchify(x, 16, "tf", out=patches))

fused = np.array([v[0] for v in rows.values()])
ops = np.array([v[1] for v in rows.values()])
gbps = lambda us: stage_bytes / us / 1e3     # noqa: E731
print("stage = RandAugment(2,9) chain -> normalise('tf') -> bf16 patch rows, batch [%d,%d,%d,3]; algorithmic bytes %.1f MB" % (B, H, W, stage_bytes / 1e6))
print("%-32s %10s %12s %10s" % ("", "us", "alg. GB/s", "of 8 TB/s"))
for label, us in (("normalise + patchify alone", t_patch), ("fused, mean of 256 pairs", fused.mean()), ("op-by-op, mean of 256 pairs", ops.mean()),
                  ("elementwise, chains by group", t_elem), ("elementwise, chains, 1 launch", t_elem_unsorted),
                  ("elementwise, dispatch x2 (r02)", t_elem_r02)):
    print("%-32s %10.1f %12.1f %9.1f%%" % (label, us, gbps(us), 100 * gbps(us) / 8000))
print("fused: min %.1f us (%s), max %.1f us (%s)" % (fused.min(), list(rows)[int(fused.argmin())], fused.max(), list(rows)[int(fused.argmax())]))
print("op-by-op: min %.1f us (%s), max %.1f us (%s)" % (ops.min(), list(rows)[int(ops.argmin())], ops.max(), list(rows)[int(ops.argmax())]))
slow = sorted(rows.items(), key=lambda kv: kv[1][0] / kv[1][1], reverse=True)[:8]
print("pairs where fusing helps least (fused us / op-by-op us):")
for k, (tf_, to_) in slow:
    print("   %-28s %8.1f / %8.1f" % (k, tf_, to_))
for title, col in (("fused stage, us (row = first op, column = second op)", 0), ("op-by-op stage, us", 1)):
    print(title)
    print("%-13s" % "" + "".join("%7s" % n[:6] for n in names))
    for a in names:
        print("%-13s" % a + "".join("%7.0f" % rows[a + ">" + b][col] for b in names))
print(json.dumps({"batch": B, "size": H, "stage_bytes": stage_bytes, "patchify_us": t_patch, "fused_mean_us": float(fused.mean()),
                  "op_by_op_mean_us": float(ops.mean()), "elementwise_us": t_elem, "elementwise_one_launch_us": t_elem_unsorted, "elementwise_dispatch_us": t_elem_r02,
                  "pairs": {k: [round(v[0], 1), round(v[1], 1)] for k, v in rows.items()}}))
