"""Hand-over from the host pipeline to the GPU: decoded images of different sizes -> one resized NHWC batch in HBM.

The reference maps `Resizing` over single images on the host and batches afterwards (training scripts; test_units/data/
test_dataset.py:176).  Here a batch of decoded images is packed back to back into one pinned staging buffer, crosses PCIe as ONE
copy on a side stream, and `chb_resize_ragged` produces the [B, OH, OW, 3] batch in a single launch (uint8 for the augmentation
kernels, or fp32 as tf.image.resize returns it).  `depth` staging slots let the copy of batch i+1 overlap the kernels of batch i.
"""
import itertools

import numpy as np
import torch

from .. import kernels as K


class _Slot:
    def __init__(self):
        self.staging = None      # pinned uint8
        self.meta = None         # pinned int64 [B * 2]: offsets | (h, w) pairs viewed as int32
        self.dev = None
        self.dev_meta = None
        self.free = None         # event: the kernels that read this slot have been issued and finished


class DeviceBatcher:
    """Iterate `(images, labels)` with images a [B, OH, OW, 3] tensor on the GPU and labels int64 [B] on the GPU.

    dataset: iterable of (uint8 [H, W, 3] array, label).  size = (OH, OW), OW % 4 == 0.  interpolation: 'bilinear' | 'nearest'.
    out_dtype: torch.uint8 (truncating cast, what the augmentation stage takes) or torch.float32."""

    def __init__(self, dataset, batch_size, size, interpolation="bilinear", out_dtype=torch.uint8, drop_remainder=False, depth=2):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceBatcher needs an MI355X (torch.cuda is not available); there is no CPU fallback")
        if interpolation not in ("bilinear", "nearest"):
            raise ValueError("unsupported interpolation %r (bilinear, nearest)" % (interpolation,))
        self.dataset, self.batch_size, self.size = dataset, int(batch_size), (int(size[0]), int(size[1]))
        self.interpolation, self.out_dtype, self.drop_remainder = interpolation, out_dtype, drop_remainder
        self.slots = [_Slot() for _ in range(max(1, depth))]
        self.copy_stream = torch.cuda.Stream()

    @staticmethod
    def pack(images):
        """Host-side layout of a ragged batch: (total bytes, int64 offsets [B], int32 hw [B, 2])."""
        hw = np.empty((len(images), 2), dtype=np.int32)
        offs = np.empty(len(images), dtype=np.int64)
        pos = 0
        for k, im in enumerate(images):
            if im.ndim != 3 or im.shape[2] != 3 or im.dtype != np.uint8 or im.shape[0] < 1 or im.shape[1] < 1:
                raise ValueError("images must be non-empty uint8 [H, W, 3] arrays")
            hw[k] = im.shape[:2]
            offs[k] = pos
            pos += im.shape[0] * im.shape[1] * 3
        return pos, offs, hw

    def __iter__(self):
        it = iter(self.dataset)
        for n in itertools.count():
            chunk = list(itertools.islice(it, self.batch_size))
            if not chunk or (self.drop_remainder and len(chunk) < self.batch_size):
                return
            images = [np.ascontiguousarray(e[0]) for e in chunk]
            labels = np.asarray([e[1] for e in chunk], dtype=np.int64)
            total, offs, hw = self.pack(images)
            b = len(images)
            slot = self.slots[n % len(self.slots)]
            if slot.free is not None:
                slot.free.synchronize()               # the batch that used this slot has been consumed
            cap = max(total, 4)
            if slot.staging is None or slot.staging.numel() < cap:
                slot.staging = torch.empty(int(cap * 1.25), dtype=torch.uint8).pin_memory()
                slot.dev = torch.empty(slot.staging.numel(), dtype=torch.uint8, device="cuda")
            if slot.meta is None or slot.meta.numel() < 3 * b:
                slot.meta = torch.empty(3 * self.batch_size, dtype=torch.int64).pin_memory()
                slot.dev_meta = torch.empty(3 * self.batch_size, dtype=torch.int64, device="cuda")
            view = slot.staging.numpy()
            for im, o in zip(images, offs):
                view[o:o + im.size] = im.reshape(-1)
            meta = slot.meta.numpy()
            meta[:b] = offs
            meta[b:2 * b].view(np.int32)[:] = hw.reshape(-1)
            meta[2 * b:3 * b] = labels
            with torch.cuda.stream(self.copy_stream):
                slot.dev[:cap].copy_(slot.staging[:cap], non_blocking=True)
                slot.dev_meta[:3 * b].copy_(slot.meta[:3 * b], non_blocking=True)
            torch.cuda.current_stream().wait_stream(self.copy_stream)
            d_offs = slot.dev_meta[:b]
            d_hw = slot.dev_meta[b:2 * b].view(torch.int32).view(b, 2)
            out = K.resize_ragged(slot.dev[:cap], d_offs, d_hw, self.size[0], self.size[1], self.interpolation, self.out_dtype)
            d_labels = slot.dev_meta[2 * b:3 * b].clone()
            slot.free = torch.cuda.Event()
            slot.free.record()
            yield out, d_labels
