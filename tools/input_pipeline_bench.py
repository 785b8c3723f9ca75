"""Input pipeline throughput: files -> decode (host threads, PIL) -> packed pinned buffer -> one H2D copy -> chb_resize_ragged.

    python tools/input_pipeline_bench.py [n_images] [batch] [workers]

Writes a synthetic image-folder dataset (JPEG, 500x375 / 375x500 / 640x480 photographs-sized, smooth random content) under
build/, then reports (a) decode-only images/s of the host pipeline, (b) images/s of the whole hand-over into a uint8
[B,224,224,3] batch in HBM, (c) the ragged resize kernel alone (HIP events, graph-free: it is ~100 us) with its algorithmic
bytes (input pixels read once + output written once).  The train step consumes ~6.3 k images/s per GPU (bench.py).
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from chambers_amd import kernels as K
from chambers_amd.data import InterleaveImageClassDataset, match_nested_set, set_n_parallel
from chambers_amd.data.device import DeviceBatcher

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
workers = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 8)
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "synthetic_images_%d" % n_images)
n_classes = 32

if not os.path.isdir(root):
    from PIL import Image
    rng = np.random.default_rng(0)
    sizes = [(375, 500), (500, 375), (480, 640)]
    for k in range(n_images):
        d = os.path.join(root, "c%02d" % (k % n_classes))
        os.makedirs(d, exist_ok=True)
        h, w = sizes[k % 3]
        low = rng.integers(0, 256, size=(h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
        img = Image.fromarray(low).resize((w, h), Image.BILINEAR)           # smooth content: realistic JPEG entropy
        img.save(os.path.join(d, "%05d.jpg" % k), quality=90)

set_n_parallel(workers)
dirs = sorted(match_nested_set(root))
per_class = n_images // n_classes


def dataset():
    return InterleaveImageClassDataset(class_dirs=dirs, labels=list(range(len(dirs))), class_cycle_length=8, images_per_block=4,
                                       block_bound=False, shuffle=True, seed=0)


t0 = time.perf_counter()
n = sum(1 for _ in dataset())
t_dec = time.perf_counter() - t0
print("decode only      : %5d images in %.2f s = %7.0f images/s  (%d worker threads)" % (n, t_dec, n / t_dec, workers))

for _ in DeviceBatcher(dataset().take(batch), batch, (224, 224)):           # warm-up: allocations, first launch
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for x, y in DeviceBatcher(dataset(), batch, (224, 224), out_dtype=torch.uint8, depth=2):
    n += int(x.shape[0])
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("files -> HBM batch: %5d images in %.2f s = %7.0f images/s  (uint8 [%d,224,224,3], one H2D copy + one launch per batch)" % (n, t_all, n / t_all, batch))

# the kernel alone
imgs = [e[0] for e in dataset().take(batch)]
total, offs, hw = DeviceBatcher.pack(imgs)
packed = torch.as_tensor(np.concatenate([im.reshape(-1) for im in imgs]), device="cuda")
d_offs, d_hw = torch.as_tensor(offs, device="cuda"), torch.as_tensor(hw, device="cuda")
out = torch.empty((batch, 224, 224, 3), dtype=torch.uint8, device="cuda")
for dt, name in ((torch.uint8, "uint8"), (torch.float32, "fp32")):
    out = torch.empty((batch, 224, 224, 3), dtype=dt, device="cuda")
    for _ in range(3):
        K.resize_ragged(packed, d_offs, d_hw, 224, 224, "bilinear", dt, out=out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        K.resize_ragged(packed, d_offs, d_hw, 224, 224, "bilinear", dt, out=out)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    alg = total + out.numel() * out.element_size()
    print("chb_resize_ragged -> %-5s: %7.1f us per batch of %d (%.0f MB in, %.0f MB out): %6.0f GB/s algorithmic, %.0f k images/s"
          % (name, us, batch, total / 1e6, out.numel() * out.element_size() / 1e6, alg / us / 1e3, batch / us * 1e3))
