"""Host-side launch wrappers: torch tensors in, C-ABI calls out (one function per entry of
include/chambers_hip.h).  torch supplies device memory and the stream, nothing else; every
function raises if its tensors are not on the GPU (no CPU fallback on the product path).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import (EPI_DGELU, EPI_GELU, EPI_NONE, EPI_PATCH, EPI_RESID, NORM_CAFFE, NORM_TF, NORM_TORCH, OUT_BF16,  # noqa: F401
                   OUT_F32, PW_BRIGHTNESS, PW_COLOR, PW_CONTRAST, PW_INVERT, PW_POSTERIZE, PW_SOLARIZE, PW_SOLARIZE_ADD)

NORM_MODES = {"caffe": NORM_CAFFE, "tf": NORM_TF, "torch": NORM_TORCH}


def _s():
    return _lib.stream_ptr()


_COPY_STREAMS = {}


def _upload(array, device):
    """Host array -> device tensor through pinned memory on a COPY stream of its own, the current stream waiting for its event: the
    transfer runs at once - while the previous training step still computes - instead of at the point of the compute stream where it
    was issued (an in-stream copy makes the stream drain, hand over to the DMA engine and pick up again: ~0.3 ms of idle GPU per
    uploaded plan in the elementwise bench; a pageable copy would moreover make the host wait for everything queued)."""
    t = torch.from_numpy(np.ascontiguousarray(array)).pin_memory()
    device = torch.device(device)
    cur = torch.cuda.current_stream(device)
    if torch.cuda.is_current_stream_capturing():
        return t.to(device, non_blocking=True)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    cs = _COPY_STREAMS.get(key)
    if cs is None:
        cs = _COPY_STREAMS[key] = torch.cuda.Stream(device)
    with torch.cuda.stream(cs):
        d = t.to(device, non_blocking=True)
    cur.wait_event(cs.record_event())
    d.record_stream(cur)
    return d


def _u8_nhwc(x):
    _lib.require_gpu(x)
    if x.dtype != torch.uint8 or x.dim() != 4:
        raise ValueError("expected a uint8 NHWC tensor of rank 4, got %s %s" % (x.dtype, tuple(x.shape)))
    return x.contiguous()


# ------------------------------------------------------------------ augmentation
def aug_pointwise(x, op, factor=0.0, i0=0, i1=0, out=None):
    x = _u8_nhwc(x)
    if op == PW_COLOR and x.shape[-1] != 3:
        raise ValueError("Color needs 3 channels")
    out = torch.empty_like(x) if out is None else out
    _lib.call("chb_aug_pointwise", _lib.ptr(x), _lib.ptr(out), x.numel(), int(op), float(factor), int(i0), int(i1), _s())
    return out


def aug_affine(x, transform, fill=0):
    """transform: 8 floats (shared by the batch) or a float32 device tensor [B,8]."""
    x = _u8_nhwc(x)
    b, h, w, c = x.shape
    out = torch.empty_like(x)
    if isinstance(transform, torch.Tensor):
        _lib.require_gpu(transform)
        t = transform.to(torch.float32).contiguous()
        if t.numel() not in (8, 8 * b):
            raise ValueError("transforms must have 8 or B*8 elements")
        _lib.call("chb_aug_affine", _lib.ptr(x), _lib.ptr(out), b, h, w, c, None, _lib.ptr(t), int(t.numel() == 8 * b), int(fill), _s())
    else:
        t = np.ascontiguousarray(np.asarray(transform, dtype=np.float32).reshape(8))
        _lib.call("chb_aug_affine", _lib.ptr(x), _lib.ptr(out), b, h, w, c, t.ctypes.data_as(ctypes.c_void_p), None, 0, int(fill), _s())
    return out


def aug_cutout(x, centers, mask_size, value=0):
    x = _u8_nhwc(x)
    b, h, w, c = x.shape
    if int(mask_size) % 2 != 0:
        raise ValueError("mask_size should be divisible by 2")
    if not isinstance(centers, torch.Tensor):
        centers = _upload(np.asarray(centers, dtype=np.int32).reshape(b, 2), x.device)
    centers = centers.to(torch.int32).contiguous()
    _lib.require_gpu(centers)
    out = torch.empty_like(x)
    _lib.call("chb_aug_cutout", _lib.ptr(x), _lib.ptr(out), b, h, w, c, _lib.ptr(centers), int(mask_size), int(value), _s())
    return out


def aug_autocontrast(x):
    x = _u8_nhwc(x)
    b, h, w, c = x.shape
    ws = torch.empty(max(b * c * 2, 1), dtype=torch.int32, device=x.device)
    out = torch.empty_like(x)
    _lib.call("chb_aug_autocontrast", _lib.ptr(x), _lib.ptr(out), b, h, w, c, _lib.ptr(ws), _s())
    return out


def aug_equalize(x):
    x = _u8_nhwc(x)
    b, h, w, c = x.shape
    ws = torch.empty(max(b * c * 256, 1), dtype=torch.int32, device=x.device)
    out = torch.empty_like(x)
    _lib.call("chb_aug_equalize", _lib.ptr(x), _lib.ptr(out), b, h, w, c, _lib.ptr(ws), _s())
    return out


def aug_sharpness(x, factor):
    x = _u8_nhwc(x)
    b, h, w, c = x.shape
    out = torch.empty_like(x)
    _lib.call("chb_aug_sharpness", _lib.ptr(x), _lib.ptr(out), b, h, w, c, float(factor), _s())
    return out


AUG_ITEM_DTYPE = np.dtype([("op", np.int32), ("i", np.int32, (4,)), ("pad", np.int32, (3,)), ("f", np.float32, (8,))])   # 64 bytes


def aug_item(op, i=(), f=()):
    """One record of chb_aug_dispatch: op id (_lib.AUG_*), up to 4 integer and 8 float parameters."""
    rec = np.zeros((), dtype=AUG_ITEM_DTYPE)
    rec["op"] = int(op)
    for k, v in enumerate(i):
        rec["i"][k] = int(v)
    for k, v in enumerate(f):
        rec["f"][k] = np.float32(v)
    return rec


def aug_dispatch(x, items, out=None):
    """One slot of an elementwise scheme: items = numpy array [B] of AUG_ITEM_DTYPE (per-image op + parameters)."""
    x = _u8_nhwc(x)
    b, h, w, c = x.shape
    if c != 3:
        raise ValueError("per-image dispatch handles RGB batches (the schemes' InputSpec), got %d channels" % c)
    items = np.ascontiguousarray(items, dtype=AUG_ITEM_DTYPE)
    if items.shape != (b,):
        raise ValueError("expected %d op records, got %s" % (b, items.shape))
    n_stats = int(np.isin(items["op"], (_lib.AUG_AUTOCONTRAST, _lib.AUG_EQUALIZE)).sum())
    dev = _upload(items.view(np.uint8).reshape(b, 64), x.device)
    ws = torch.empty(max(b * 768, 1), dtype=torch.int32, device=x.device) if n_stats else None
    out = torch.empty_like(x) if out is None else out
    _lib.call("chb_aug_dispatch", _lib.ptr(x), _lib.ptr(out), b, h, w, _lib.ptr(dev), n_stats, _lib.ptr(ws), _s())
    return out


FUSED_OP_DTYPE = np.dtype([("op", np.int32), ("i", np.int32, (4,)), ("f", np.float32, (6,)), ("pad", np.int32)])   # 48 bytes
FUSED_MAX_OPS = 4
ITEMS_GROUPS = 6          # CHB_ITEMS_GROUPS: columns of chb_aug_items_sort's counts


class AugPlan:
    """The batch-shared decisions of one scheme call, resolved to op records: what chb_aug_fused evaluates per pixel.
    items: AUG_ITEM_DTYPE records in application order; centers: per level None or int32 [B,2] (CutOut's per-image centres)."""

    def __init__(self, items, centers=None):
        self.items = list(items)
        self.centers = list(centers) if centers is not None else [None] * len(self.items)
        if len(self.centers) != len(self.items):
            raise ValueError("one centres entry per op")

    def __len__(self):
        return len(self.items)

    @property
    def n_tables(self):
        return sum(int(it["op"]) in (_lib.AUG_AUTOCONTRAST, _lib.AUG_EQUALIZE) for it in self.items)


def aug_fused(x, plan, patch=None, out=None, scratch=True):
    """One launch for a whole batch-shared op chain (plus one histogram pass per AutoContrast / Equalize in it).
    patch=None: uint8 NHWC result;  patch=P: bf16 patch rows of the "tf"-normalised result (normalize_patchify fused in).
    scratch=True lets a chain be cut at its Sharpness ops (chb_aug_fused's scratch argument)."""
    x = _u8_nhwc(x)
    b, h, w, c = x.shape
    if c != 3:
        raise ValueError("the fused scheme stage handles RGB batches (the schemes' InputSpec), got %d channels" % c)
    n = len(plan)
    if not 1 <= n <= FUSED_MAX_OPS:
        raise ValueError("a fused chain holds 1..%d ops, got %d" % (FUSED_MAX_OPS, n))
    recs = np.zeros(n, dtype=FUSED_OP_DTYPE)
    cptr = (ctypes.c_void_p * n)()
    keep = []
    for l, it in enumerate(plan.items):
        recs[l]["op"] = it["op"]
        recs[l]["i"] = it["i"]
        recs[l]["f"] = it["f"][:6]
        if int(it["op"]) == _lib.AUG_CUTOUT:
            cen = plan.centers[l]
            if isinstance(cen, torch.Tensor):        # resident decisions
                _lib.require_gpu(cen)
                dev = cen.to(torch.int32).contiguous()
            else:
                cen = np.ascontiguousarray(cen, dtype=np.int32)
                dev = None
            if tuple(cen.shape) != (b, 2):
                raise ValueError("CutOut wants one (cy, cx) per image: expected %s, got %s" % ((b, 2), tuple(cen.shape)))
            if dev is None:
                dev = _upload(cen, x.device)
            keep.append(dev)
            cptr[l] = dev.data_ptr()
    nt = plan.n_tables
    ws = torch.empty(_lib.aug_fused_workspace_ints(b, h, w, nt), dtype=torch.int32, device=x.device) if nt and b else None
    if patch is None:
        out = torch.empty_like(x) if out is None else out
        if out.shape != x.shape or out.dtype != torch.uint8 or not out.is_contiguous():
            raise ValueError("out must be a contiguous uint8 tensor shaped like the input")
    else:
        rows = b * (h // patch) * (w // patch)
        if out is None:
            out = torch.empty((rows, patch * patch * 3), dtype=torch.bfloat16, device=x.device)
        if out.dtype != torch.bfloat16 or out.numel() < rows * patch * patch * 3 or not out.is_contiguous():
            raise ValueError("out must be a contiguous bf16 buffer of at least %d patch rows" % rows)
    ops = [int(it["op"]) for it in plan.items]
    cut = _lib.AUG_SHARPNESS in ops and (n > 1 or patch is not None)
    scratch = torch.empty((2,) + tuple(x.shape), dtype=torch.uint8, device=x.device) if (cut and scratch) else None
    _lib.call("chb_aug_fused", _lib.ptr(x), _lib.ptr(out), b, h, w, n, recs.ctypes.data, ctypes.cast(cptr, ctypes.c_void_p),
              _lib.ptr(ws), _lib.ptr(scratch), 0 if patch is None else int(patch), _s())
    return out


class AugItemsPlan:
    """Per-image chains of an elementwise scheme: `items` = numpy [n_slots, B] of AUG_ITEM_DTYPE (what the transforms' dispatch_item
    returns).  ViTEngine.forward(..., augment=AugItemsPlan) evaluates them inside its normalise + patchify pass.  The device copy of
    the records is made on first use and kept (a plan replayed from a HIP graph uploads nothing)."""

    def __init__(self, items):
        self.items = np.ascontiguousarray(items, dtype=AUG_ITEM_DTYPE)
        if self.items.ndim != 2:
            raise ValueError("expected [n_slots, B] op records, got shape %s" % (self.items.shape,))
        self._resident = None

    def __len__(self):
        return int(self.items.shape[0])

    def resident(self, device, h=None, w=None):
        """(device records [n*B, 48] uint8, per-level cutout-centre tensors or None, table-level bit mask, group order, group counts).
        With the image size given the images are sorted by what their chains need (chb_aug_items_sort: device int32 [(1+n)*B] order
        + host int32 [(1+n)*6] counts, for chb_aug_fused_items_sorted); without it the last two are None."""
        key = (device, h, w)
        if self._resident is None or self._resident[0] != key:
            items = self.items
            n, b = items.shape
            recs = np.zeros((n, b), dtype=FUSED_OP_DTYPE)
            recs["op"] = items["op"]
            recs["i"] = items["i"]
            recs["f"] = items["f"][..., :6]
            centers, tables = [], 0
            for l in range(n):
                ops = items[l]["op"]
                if np.isin(ops, (_lib.AUG_AUTOCONTRAST, _lib.AUG_EQUALIZE)).any():
                    tables |= 1 << l
                # the dispatch record carries a CutOut's centre in i0, i1; the chain evaluators read a [B,2] table
                centers.append(_upload(np.ascontiguousarray(items[l]["i"][:, :2], dtype=np.int32), device) if (ops == _lib.AUG_CUTOUT).any() else None)
            order = counts = None
            if h is not None and b and 1 <= n <= FUSED_MAX_OPS:
                order_host = np.zeros(((1 + n), b), dtype=np.int32)
                counts = np.zeros((1 + n) * ITEMS_GROUPS, dtype=np.int32)
                _lib.call("chb_aug_items_sort", recs.ctypes.data, b, int(h), int(w), n, order_host.ctypes.data, counts.ctypes.data)
                order = _upload(order_host.reshape(-1), device)
            self._resident = (key, (_upload(recs.view(np.uint8).reshape(n * b, FUSED_OP_DTYPE.itemsize), device), centers, tables, order, counts))
        return self._resident[1]


def aug_fused_items(x, items, patch=None, out=None):
    """The whole per-image chains, the images sorted by what their chain needs and one launch per group (chb_aug_fused_items_sorted;
    + histogram passes and a table launch per slot in which some image drew AutoContrast / Equalize).  items: [n_slots, B] AUG_ITEM_DTYPE or an AugItemsPlan; patch as for aug_fused."""
    x = _u8_nhwc(x)
    b, h, w, c = x.shape
    if c != 3:
        raise ValueError("per-image chains handle RGB batches (the schemes' InputSpec), got %d channels" % c)
    plan = items if isinstance(items, AugItemsPlan) else AugItemsPlan(items)
    n = len(plan)
    if plan.items.shape[1] != b or not 1 <= n <= FUSED_MAX_OPS:
        raise ValueError("expected [1..%d, %d] op records, got %s" % (FUSED_MAX_OPS, b, plan.items.shape))
    dev_items, centers, tables, order, counts = plan.resident(x.device, h, w)
    cptr = (ctypes.c_void_p * n)()
    for l, cen in enumerate(centers):
        if cen is not None:
            cptr[l] = cen.data_ptr()
    nt = bin(tables).count("1")
    ws = torch.empty(_lib.aug_fused_workspace_ints(b, h, w, nt), dtype=torch.int32, device=x.device) if nt and b else None
    if patch is None:
        out = torch.empty_like(x) if out is None else out
        if out.shape != x.shape or out.dtype != torch.uint8 or not out.is_contiguous():
            raise ValueError("out must be a contiguous uint8 tensor shaped like the input")
    else:
        rows = b * (h // patch) * (w // patch)
        if out is None:
            out = torch.empty((rows, patch * patch * 3), dtype=torch.bfloat16, device=x.device)
        if out.dtype != torch.bfloat16 or out.numel() < rows * patch * patch * 3 or not out.is_contiguous():
            raise ValueError("out must be a contiguous bf16 buffer of at least %d patch rows" % rows)
    if b == 0:
        return out
    _lib.call("chb_aug_fused_items_sorted", _lib.ptr(x), _lib.ptr(out), b, h, w, n, _lib.ptr(dev_items), ctypes.cast(cptr, ctypes.c_void_p), tables,
              _lib.ptr(ws), 0 if patch is None else int(patch), _lib.ptr(order), counts.ctypes.data, _s())
    return out


def concat_batch(parts):
    """Batch-axis concatenation of same-shaped image tensors by device-to-device copies into one allocation (the
    image-by-image route of elementwise schemes with user-supplied transforms)."""
    if not parts:
        raise ValueError("nothing to concatenate")
    total = sum(int(p.shape[0]) for p in parts)
    out = torch.empty((total,) + tuple(parts[0].shape[1:]), dtype=parts[0].dtype, device=parts[0].device)
    at = 0
    for p in parts:
        out[at:at + p.shape[0]].copy_(p)
        at += p.shape[0]
    return out


def normalize(x, mode):
    _lib.require_gpu(x)
    if mode not in NORM_MODES:
        raise ValueError("Unknown mode " + str(mode))
    x = x.contiguous()
    c = x.shape[-1]
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    n_pixels = x.numel() // c
    if x.dtype == torch.uint8:
        _lib.call("chb_normalize_u8", _lib.ptr(x), _lib.ptr(out), n_pixels, c, NORM_MODES[mode], _s())
    else:
        x = x.to(torch.float32)
        _lib.call("chb_normalize_f32", _lib.ptr(x), _lib.ptr(out), n_pixels, c, NORM_MODES[mode], _s())
    return out


def normalize_patchify(x, patch, mode="tf", out=None):
    x = _u8_nhwc(x)
    b, h, w, c = x.shape
    if c != 3:
        raise ValueError("patchify expects RGB input")
    rows = b * (h // patch) * (w // patch)
    if out is None:
        out = torch.empty((rows, patch * patch * 3), dtype=torch.bfloat16, device=x.device)
    _lib.call("chb_normalize_patchify_bf16", _lib.ptr(x), _lib.ptr(out), b, h, w, int(patch), NORM_MODES[mode], _s())
    return out


def patchify_f32(x, patch, out=None):
    _lib.require_gpu(x)
    if x.dtype != torch.float32 or x.dim() != 4 or x.shape[-1] != 3:
        raise ValueError("expected a float32 NHWC RGB tensor")
    x = x.contiguous()
    b, h, w, _ = x.shape
    rows = b * (h // patch) * (w // patch)
    if out is None:
        out = torch.empty((rows, patch * patch * 3), dtype=torch.bfloat16, device=x.device)
    _lib.call("chb_patchify_f32_bf16", _lib.ptr(x), _lib.ptr(out), b, h, w, int(patch), _s())
    return out


# ------------------------------------------------------------------ dropout mask
def dropout_mask(n, rate, key, device="cuda"):
    out = torch.empty(n, dtype=torch.uint8, device=device)
    _lib.call("chb_dropout_mask", _lib.ptr(out), n, float(rate), ctypes.c_uint32(int(key)), _s())
    return out


# ------------------------------------------------------------------ input side
def _nhwc(x):
    if x.dim() != 4:
        raise ValueError("expected a 4-D NHWC batch, got shape %s" % (tuple(x.shape),))
    if x.dtype not in (torch.uint8, torch.float32):
        raise ValueError("expected uint8 or float32 images, got %s" % (x.dtype,))
    _lib.require_gpu(x)
    return x.contiguous()


def resize(x, out_h, out_w, method="bilinear"):
    """tf.image.resize (half-pixel centres, no antialias): bilinear -> float32, nearest -> dtype of x."""
    x = _nhwc(x)
    if method not in ("bilinear", "nearest"):
        raise ValueError("unsupported interpolation %r (bilinear, nearest)" % (method,))
    b, h, w, c = x.shape
    out = torch.empty((b, int(out_h), int(out_w), c), dtype=torch.float32 if method == "bilinear" else x.dtype, device=x.device)
    _lib.call("chb_resize", _lib.ptr(x), 0 if x.dtype == torch.uint8 else 1, _lib.ptr(out), b, h, w, c, int(out_h), int(out_w),
              0 if method == "bilinear" else 1, _s())
    return out


def resize_ragged(packed, offsets, hw, out_h, out_w, method="bilinear", out_dtype=torch.float32, out=None):
    """B RGB uint8 images of different sizes, packed back to back (`packed` uint8 [bytes] on the GPU, `offsets` int64 [B],
    `hw` int32 [B,2]) -> one [B,out_h,out_w,3] batch in a single launch; fp32 (tf.image.resize) or uint8 (its truncating cast)."""
    _lib.require_gpu(packed, offsets, hw)
    if method not in ("bilinear", "nearest"):
        raise ValueError("unsupported interpolation %r (bilinear, nearest)" % (method,))
    if packed.dtype != torch.uint8 or offsets.dtype != torch.int64 or hw.dtype != torch.int32:
        raise ValueError("packed uint8, offsets int64, hw int32 expected")
    if out_dtype not in (torch.float32, torch.uint8):
        raise ValueError("out_dtype must be float32 or uint8")
    b = int(offsets.numel())
    if tuple(hw.shape) != (b, 2):
        raise ValueError("hw must be [B, 2] (height, width)")
    if out is None:
        out = torch.empty((b, int(out_h), int(out_w), 3), dtype=out_dtype, device=packed.device)
    _lib.call("chb_resize_ragged", _lib.ptr(packed), int(packed.numel()), _lib.ptr(offsets), _lib.ptr(hw), b, _lib.ptr(out),
              0 if out.dtype == torch.uint8 else 1, int(out_h), int(out_w), 0 if method == "bilinear" else 1, _s())
    return out


def crop_flip(x, out_h, out_w, offsets=None, flips=None):
    """Window gather with optional per-image flips.  offsets: (y, x) tuple for the whole batch, or an int32 [B,2] array /
    tensor; flips: uint8 [B] (bit 0 left-right, bit 1 up-down) or None."""
    x = _nhwc(x)
    b, h, w, c = x.shape
    out = torch.empty((b, int(out_h), int(out_w), c), dtype=x.dtype, device=x.device)
    oy0 = ox0 = 0
    off_t = None
    per_image = 0
    if offsets is not None:
        arr = offsets if isinstance(offsets, torch.Tensor) else np.asarray(offsets, dtype=np.int32)
        if tuple(arr.shape) == (2,):
            oy0, ox0 = int(arr[0]), int(arr[1])
        else:
            if isinstance(arr, np.ndarray):
                if arr.shape != (b, 2) or arr.min() < 0 or (arr[:, 0] + out_h).max() > h or (arr[:, 1] + out_w).max() > w:
                    raise ValueError("crop windows must lie inside the image")
                arr = torch.as_tensor(arr, device=x.device)
            off_t = arr.to(torch.int32).contiguous()
            per_image = 1
    fl_t = None
    if flips is not None:
        fl_t = (flips if isinstance(flips, torch.Tensor) else torch.as_tensor(np.asarray(flips, dtype=np.uint8), device=x.device)).to(torch.uint8).contiguous()
    _lib.call("chb_crop_flip", _lib.ptr(x), _lib.ptr(out), b, h, w, c * x.element_size(), int(out_h), int(out_w),
              _lib.ptr(off_t) if off_t is not None else None, per_image, oy0, ox0, _lib.ptr(fl_t) if fl_t is not None else None, _s())
    return out


def rescale(x, scale, offset=0.0):
    if x.dtype not in (torch.uint8, torch.float32):
        raise ValueError("expected uint8 or float32, got %s" % (x.dtype,))
    _lib.require_gpu(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.call("chb_rescale", _lib.ptr(x), 0 if x.dtype == torch.uint8 else 1, _lib.ptr(out), x.numel(), float(scale), float(offset), _s())
    return out


# ------------------------------------------------------------------ ViT block
def gemm_nt(a, b, out, m=None, bias=None, epilogue=EPI_NONE, aux=None, resid=None, period=0, drop_rate=0.0, drop_key=0, colsum=None):
    """out[M,N] = epi(a[M,K] . b[N,K]^T); a, b bf16 2-D (row stride taken from the tensors)."""
    _lib.require_gpu(a, b, out, bias, aux, resid, colsum)
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
        raise ValueError("gemm operands must be bfloat16")
    m = a.shape[0] if m is None else m
    n, k = b.shape[0], b.shape[1]
    if a.shape[1] != k:
        raise ValueError("inner dimensions differ: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    out_dtype = OUT_F32 if out.dtype == torch.float32 else OUT_BF16
    _lib.call("chb_gemm_nt", _lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), _lib.ptr(out), out.stride(0), int(m), int(n), int(k),
              _lib.ptr(bias), int(epilogue), out_dtype, _lib.ptr(aux), aux.stride(0) if aux is not None else 0, _lib.ptr(resid),
              resid.stride(0) if resid is not None else 0, int(period), float(drop_rate), ctypes.c_uint32(int(drop_key)), _lib.ptr(colsum), _s())
    return out


def gemm_tn(x, dy, dw, m=None, ws=None, fold=True, colsum=None):
    """dw[Kd,Nd] += x[M,Kd]^T . dy[M,Nd] (fp32 accumulate).  ws: optional fp32 scratch tensor for the split-K partial planes
    (tn_workspace_elems(Kd, Nd) elements suffice); without it the partials meet in fp32 atomics.  fold=False leaves the planes
    in ws until gemm_tn_fold (same arguments).  colsum (fp32 [Nd]): += column sums of dy (bias gradient), inside the GEMM."""
    _lib.require_gpu(x, dy, dw, ws, colsum)
    m = x.shape[0] if m is None else m
    if ws is None and colsum is None:
        _lib.call("chb_gemm_tn", _lib.ptr(x), x.stride(0), _lib.ptr(dy), dy.stride(0), _lib.ptr(dw), dw.stride(0), int(m), x.shape[1],
                  dy.shape[1], _s())
    else:
        if ws is not None and ws.dtype != torch.float32:
            raise ValueError("ws must be fp32")
        if colsum is not None and (colsum.dtype != torch.float32 or colsum.numel() < dy.shape[1]):
            raise ValueError("colsum must be fp32 [Nd]")
        _lib.call("chb_gemm_tn_ws", _lib.ptr(x), x.stride(0), _lib.ptr(dy), dy.stride(0), _lib.ptr(dw), dw.stride(0), int(m), x.shape[1],
                  dy.shape[1], _lib.ptr(ws), (int(ws.numel()) * 4) if ws is not None else 0, 1 if fold else 0, _lib.ptr(colsum), _s())
    return dw


def gemm_tn_fold(x, dy, dw, m=None, ws=None):
    """Second half of gemm_tn(..., ws=ws, fold=False): dw += sum of the partial planes (no-op if that call used atomics)."""
    if ws is None:
        return dw
    _lib.require_gpu(dw, ws)
    m = x.shape[0] if m is None else m
    _lib.call("chb_gemm_tn_fold", _lib.ptr(ws), int(ws.numel()) * 4, _lib.ptr(dw), dw.stride(0), int(m), x.shape[1], dy.shape[1], _s())
    return dw


def tn_workspace_elems(kd, nd, n_cus=256):
    """fp32 elements that always suffice as gemm_tn scratch for a [kd, nd] gradient: one plane per split, splits <= CUs / tiles."""
    tiles = ((kd + 255) // 256) * ((nd + 255) // 256)
    return max(1, n_cus // tiles) * kd * nd


def layernorm_fwd(x, x_stride, gamma, beta, y, mean, rstd, m, d, eps):
    _lib.require_gpu(x, gamma, beta, y, mean, rstd)
    _lib.call("chb_layernorm_fwd", _lib.ptr(x), int(x_stride), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd),
              int(m), int(d), float(eps), _s())
    return y


def layernorm_bwd(dy, x, x_stride, mean, rstd, gamma, dx, dx_stride, accumulate, dgamma, dbeta, m, d, dz=None, dz_colsum=None,
                  drop_rate=0.0, drop_key=0, zero_gaps=False):
    _lib.require_gpu(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, dz, dz_colsum)
    if dz is not None and int(dx_stride) != int(d) and not (zero_gaps and int(dx_stride) % int(d) == 0):
        raise ValueError("the fused dropout-backward tail needs compact rows, or strided rows whose gaps this launch zero-fills")
    _lib.call("chb_layernorm_bwd", _lib.ptr(dy), _lib.ptr(x), int(x_stride), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(dx),
              int(dx_stride), int(bool(accumulate)), _lib.ptr(dgamma), _lib.ptr(dbeta), int(m), int(d), _lib.ptr(dz), _lib.ptr(dz_colsum),
              float(drop_rate), ctypes.c_uint32(int(drop_key)), int(bool(zero_gaps)), _s())
    return dx


def attention_drop_bits(b, n, h, device="cuda"):
    """Buffer for the keep bits of one attention site (forward writes, backward reads): uint32 [B*H*N*8] (int32 storage)."""
    return torch.empty(int(b) * int(h) * int(n) * 8, dtype=torch.int32, device=device)


def attention_fwd(qkv, o, lse, b, n, h, hd, drop_rate=0.0, drop_key=0, drop_bits=None):
    _lib.require_gpu(qkv, o, lse, drop_bits)
    if drop_bits is not None and (drop_bits.dtype != torch.int32 or drop_bits.numel() < int(b) * int(h) * int(n) * 8):
        raise ValueError("drop_bits must be int32 with at least B*H*N*8 elements")
    _lib.call("chb_attention_fwd", _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), int(b), int(n), int(h), int(hd), float(drop_rate),
              ctypes.c_uint32(int(drop_key)), _lib.ptr(drop_bits), _s())
    return o


def attention_bwd(qkv, o, d_o, lse, dqkv, b, n, h, hd, drop_rate=0.0, drop_key=0, dbias=None, dbias_ws=None, drop_bits=None):
    """dbias (fp32 [3*h*hd]): += column sums of dqkv; dbias_ws (fp32 [b, 3*h*hd] scratch): per-head sums go there with plain
    stores and one small launch folds them (n <= 224), instead of one atomic per head and column."""
    _lib.require_gpu(qkv, o, d_o, lse, dqkv, dbias, dbias_ws, drop_bits)
    if drop_bits is not None and (drop_bits.dtype != torch.int32 or drop_bits.numel() < int(b) * int(h) * int(n) * 8):
        raise ValueError("drop_bits must be int32 with at least B*H*N*8 elements")
    if dbias is not None and dbias_ws is None:
        dbias_ws = torch.empty(int(b) * 3 * int(h) * int(hd), dtype=torch.float32, device=dbias.device)
    if dbias_ws is not None and (dbias_ws.dtype != torch.float32 or dbias_ws.numel() < int(b) * 3 * int(h) * int(hd)):
        raise ValueError("dbias_ws must be fp32 with at least b * 3 * h * hd elements")
    _lib.call("chb_attention_bwd", _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(d_o), _lib.ptr(lse), _lib.ptr(dqkv), int(b), int(n), int(h), int(hd),
              float(drop_rate), ctypes.c_uint32(int(drop_key)), _lib.ptr(dbias), _lib.ptr(dbias_ws), _lib.ptr(drop_bits), _s())
    return dqkv


def cls_row(x, cls, pos, b, n, d, drop_rate=0.0, drop_key=0):
    _lib.require_gpu(x, cls, pos)
    _lib.call("chb_cls_row", _lib.ptr(x), _lib.ptr(cls), _lib.ptr(pos), int(b), int(n), int(d), float(drop_rate), ctypes.c_uint32(int(drop_key)), _s())


def token_row(x, tok, pos, b, n, d, row, drop_rate=0.0, drop_key=0):
    """x[b, row, :] = dropout(tok + pos[row]) — row 0 class token, row 1 distillation token."""
    _lib.require_gpu(x, tok, pos)
    _lib.call("chb_token_row", _lib.ptr(x), _lib.ptr(tok), _lib.ptr(pos), int(b), int(n), int(d), int(row), float(drop_rate),
              ctypes.c_uint32(int(drop_key)), _s())


def embed_bwd(dx, dpatch, dpos, dtok, b, n, d, drop_rate=0.0, drop_key=0, n_special=1):
    """dtok: fp32 [n_special, d] (or [d] for the class token alone)."""
    _lib.require_gpu(dx, dpatch, dpos, dtok)
    _lib.call("chb_embed_bwd_tokens", _lib.ptr(dx), _lib.ptr(dpatch), _lib.ptr(dpos), _lib.ptr(dtok), int(b), int(n), int(d), int(n_special),
              float(drop_rate), ctypes.c_uint32(int(drop_key)), _s())


def dropout_bwd(dy, dz, m, n, drop_rate=0.0, drop_key=0):
    _lib.require_gpu(dy, dz)
    _lib.call("chb_dropout_bwd_bf16", _lib.ptr(dy), dy.stride(0), _lib.ptr(dz), int(m), int(n), float(drop_rate), ctypes.c_uint32(int(drop_key)), _s())
    return dz


POOL_MODES = {"avg": 0, "max": 1, "sum": 2}


def pool_tokens(h, out, argmax, b, n, d, mode):
    """pooling over the patch tokens 1..n-1 of h bf16 [b*n, d] (vision_transformer.py:172-181)."""
    _lib.require_gpu(h, out)
    _lib.call("chb_pool_tokens", _lib.ptr(h), _lib.ptr(out), _lib.ptr(argmax) if argmax is not None else None, int(b), int(n), int(d),
              POOL_MODES[mode], _s())
    return out


def pool_tokens_bwd(dout, argmax, dh, b, n, d, mode):
    _lib.require_gpu(dout, dh)
    _lib.call("chb_pool_tokens_bwd", _lib.ptr(dout), _lib.ptr(argmax) if argmax is not None else None, _lib.ptr(dh), int(b), int(n), int(d),
              POOL_MODES[mode], _s())
    return dh


def tanh_fwd(z, y_bf16=None):
    """z fp32 (contiguous) -> tanh in place, optional bf16 copy."""
    _lib.require_gpu(z)
    _lib.call("chb_tanh_fwd", _lib.ptr(z), _lib.ptr(y_bf16) if y_bf16 is not None else None, z.numel(), _s())
    return z


def tanh_bwd(dy, y, dz_bf16):
    _lib.require_gpu(dy, y, dz_bf16)
    _lib.call("chb_tanh_bwd", _lib.ptr(dy), _lib.ptr(y), _lib.ptr(dz_bf16), dy.numel(), _s())
    return dz_bf16


def colsum(x, out, m=None):
    _lib.require_gpu(x, out)
    m = x.shape[0] if m is None else m
    _lib.call("chb_colsum_bf16", _lib.ptr(x), x.stride(0), _lib.ptr(out), int(m), x.shape[1], _s())
    return out


def softmax_ce(logits, labels, loss, dlogits, classes, grad_scale):
    _lib.require_gpu(logits, labels, loss, dlogits)
    _lib.call("chb_softmax_ce", _lib.ptr(logits), logits.stride(0), _lib.ptr(labels), _lib.ptr(loss), _lib.ptr(dlogits),
              dlogits.stride(0) if dlogits is not None else 0, logits.shape[0], int(classes), float(grad_scale), _s())
    return loss


def cast_transpose(src, dst, dst_t, desc, n_desc, max_tiles):
    _lib.require_gpu(src, dst, dst_t, desc)
    _lib.call("chb_cast_transpose", _lib.ptr(src), _lib.ptr(dst), _lib.ptr(dst_t), _lib.ptr(desc), int(n_desc), int(max_tiles), _s())


def adamw(p, g, m, v, flags, lr_t, beta1, beta2, eps, weight_decay, grad_scale=1.0, zero_grad=False):
    _lib.require_gpu(p, g, m, v, flags)
    _lib.call("chb_adamw", _lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), _lib.ptr(flags), p.numel(), float(lr_t), float(beta1), float(beta2),
              float(eps), float(weight_decay), float(grad_scale), int(bool(zero_grad)), _s())


def zero_f32(x):
    _lib.require_gpu(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("zero_f32 expects a contiguous float32 tensor")
    _lib.call("chb_zero_f32", _lib.ptr(x), x.numel(), _s())
    return x


# ------------------------------------------------------------------ small tensor utilities (stand-alone layers)
def add_f32(a, b, out=None):
    _lib.require_gpu(a, b)
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != b.shape:
        raise ValueError("add_f32 takes two fp32 tensors of one shape")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a) if out is None else out
    _lib.call("chb_add_f32", _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.numel(), _s())
    return out


def axpby_f32(a, alpha, b=None, beta=0.0):
    """alpha * a (+ beta * b), fp32, contiguous tensors of one shape."""
    _lib.require_gpu(a, b)
    if a.dtype != torch.float32 or (b is not None and (b.dtype != torch.float32 or b.shape != a.shape)):
        raise ValueError("axpby_f32 takes fp32 tensors of one shape")
    a = a.contiguous()
    b = b.contiguous() if b is not None else None
    out = torch.empty_like(a)
    _lib.call("chb_axpby_f32", _lib.ptr(a), float(alpha), _lib.ptr(b), float(beta), _lib.ptr(out), a.numel(), _s())
    return out


def add_rows_bf16(x, y):
    """x += y for 2-D bf16 views whose rows are contiguous (any row stride): fp32 sum, one rounding."""
    _lib.require_gpu(x, y)
    if x.dtype != torch.bfloat16 or y.dtype != torch.bfloat16 or x.dim() != 2 or x.shape != y.shape or x.stride(1) != 1 or y.stride(1) != 1:
        raise ValueError("add_rows_bf16 takes two 2-D bf16 views of one shape with contiguous rows")
    _lib.call("chb_add_rows_bf16", _lib.ptr(x), x.stride(0), _lib.ptr(y), y.stride(0), x.shape[0], x.shape[1], _s())
    return x


def store_rows(dst, src):
    """dst[...] = src for a 2-D destination view with contiguous rows (any row stride): a cast kernel when the dtypes differ
    (fp32 -> bf16, round to nearest even; bf16 -> fp32), then a strided row copy - no torch arithmetic."""
    _lib.require_gpu(dst, src)
    if dst.dim() != 2 or dst.stride(1) != 1 or tuple(src.shape) != tuple(dst.shape):
        raise ValueError("store_rows: dst must be a 2-D view with contiguous rows and src must have its shape; got %s <- %s" % (tuple(dst.shape), tuple(src.shape)))
    if src.dtype != dst.dtype:
        src = cast_bf16(src) if dst.dtype == torch.bfloat16 else cast_f32(src)
    if src.stride(1) != 1:
        src = src.contiguous()
    es = dst.element_size()
    if (dst.shape[1] * es) % 4 or (dst.stride(0) * es) % 4 or (src.stride(0) * es) % 4:
        raise ValueError("store_rows: row bytes and strides must be multiples of 4")
    _lib.call("chb_copy_rows", _lib.ptr(src), src.stride(0) * es, _lib.ptr(dst), dst.stride(0) * es, dst.shape[0], dst.shape[1] * es, _s())
    return dst


def cast_bf16(x, out=None):
    """fp32 -> bf16 (round to nearest even), a bf16 tensor is returned as it is.  out: a contiguous bf16 buffer with at least
    x.numel() elements whose head receives the result (rows of a zero-padded operand); a bf16 x is copied there."""
    if x.dtype == torch.bfloat16 and out is None:
        return x
    _lib.require_gpu(x, out)
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("cast_bf16 takes fp32 or bf16, got %s" % x.dtype)
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    elif out.dtype != torch.bfloat16 or not out.is_contiguous() or out.numel() < x.numel():
        raise ValueError("cast_bf16: out must be a contiguous bf16 buffer of at least x.numel() elements")
    if x.dtype == torch.bfloat16:
        out.view(-1)[:x.numel()].copy_(x.view(-1))          # same dtype: a device-to-device copy
        return out
    _lib.call("chb_cast_f32_bf16", _lib.ptr(x), _lib.ptr(out), x.numel(), _s())
    return out


def cast_f32(x):
    if x.dtype == torch.float32:
        return x
    _lib.require_gpu(x)
    if x.dtype != torch.bfloat16:
        raise ValueError("cast_f32 takes bf16 or fp32, got %s" % x.dtype)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.call("chb_cast_bf16_f32", _lib.ptr(x), _lib.ptr(out), x.numel(), _s())
    return out


def concat_axis1(parts):
    """[B, n_i, D] tensors of one dtype -> [B, sum n_i, D] (tf.concat(axis=1)) by strided row copies."""
    _lib.require_gpu(*parts)
    b, d = parts[0].shape[0], parts[0].shape[2]
    es = parts[0].element_size()
    if any(p.dim() != 3 or p.shape[0] != b or p.shape[2] != d or p.dtype != parts[0].dtype for p in parts) or (d * es) % 4:
        raise ValueError("concat_axis1 takes [B, n_i, D] tensors of one dtype with D * itemsize % 4 == 0")
    n = sum(int(p.shape[1]) for p in parts)
    out = torch.empty((b, n, d), dtype=parts[0].dtype, device=parts[0].device)
    at = 0
    for p in parts:
        p = p.contiguous()
        ni = int(p.shape[1])
        dst = ctypes.c_void_p(out.data_ptr() + at * d * es)
        _lib.call("chb_copy_rows", _lib.ptr(p), ni * d * es, dst, n * d * es, b, ni * d * es, _s())
        at += ni
    return out


def softmax_rows(x):
    """Softmax over the last axis of an fp32 tensor."""
    _lib.require_gpu(x)
    if x.dtype != torch.float32:
        raise ValueError("softmax_rows takes fp32")
    x = x.contiguous()
    cols = int(x.shape[-1])
    rows = x.numel() // max(cols, 1)
    out = torch.empty_like(x)
    _lib.call("chb_softmax_f32", _lib.ptr(x), cols, _lib.ptr(out), cols, rows, cols, _s())
    return out


# ------------------------------------------------------------------ pieces of the trainable stand-alone layers
def gelu_f32(x, approximate=False, want_derivative=False):
    """y = gelu(x) (fp32), optionally with dy/dx (activations.py:5-56: exact-erf or tanh form)."""
    _lib.require_gpu(x)
    if x.dtype != torch.float32:
        raise ValueError("gelu_f32 takes fp32")
    x = x.contiguous()
    y = torch.empty_like(x)
    d = torch.empty_like(x) if want_derivative else None
    _lib.call("chb_gelu_f32", _lib.ptr(x), _lib.ptr(y), _lib.ptr(d), x.numel(), int(bool(approximate)), _s())
    return (y, d) if want_derivative else y


def mul_f32(a, b):
    _lib.require_gpu(a, b)
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != b.shape:
        raise ValueError("mul_f32 takes two fp32 tensors of one shape")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    _lib.call("chb_mul_f32", _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.numel(), _s())
    return out


def scale_by_bf16(dy, aux, out=None):
    """bf16(dy * aux): dy fp32 or bf16, aux bf16 of the same shape (out: optional buffer whose head receives the result)."""
    _lib.require_gpu(dy, aux)
    if aux.dtype != torch.bfloat16 or dy.shape != aux.shape or dy.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("scale_by_bf16 takes dy (fp32 | bf16) and a bf16 aux of the same shape")
    dy, aux = dy.contiguous(), aux.contiguous()
    if out is None:
        out = torch.empty(dy.shape, dtype=torch.bfloat16, device=dy.device)
    elif out.dtype != torch.bfloat16 or not out.is_contiguous() or out.numel() < dy.numel():
        raise ValueError("scale_by_bf16: out must be a contiguous bf16 buffer of at least dy.numel() elements")
    _lib.call("chb_scale_by_bf16", _lib.ptr(dy), OUT_F32 if dy.dtype == torch.float32 else OUT_BF16, _lib.ptr(aux), _lib.ptr(out), dy.numel(), _s())
    return out


def dropout_f32(x, rate, key):
    """x * keep / (1 - rate) with the flat-index keep mask of chb_dropout_mask (forward and backward of a Dropout layer)."""
    _lib.require_gpu(x)
    if x.dtype != torch.float32:
        raise ValueError("dropout_f32 takes fp32")
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.call("chb_dropout_f32", _lib.ptr(x), _lib.ptr(out), x.numel(), float(rate), ctypes.c_uint32(int(key)), _s())
    return out


def add_rows_f32(x, table):
    """x [..., *table.shape] + table broadcast over the leading axes (fp32)."""
    _lib.require_gpu(x, table)
    if x.dtype != torch.float32 or table.dtype != torch.float32 or x.numel() % max(table.numel(), 1):
        raise ValueError("add_rows_f32 takes fp32 x whose trailing axes are the table's")
    x, table = x.contiguous(), table.contiguous()
    out = torch.empty_like(x)
    _lib.call("chb_add_rows_f32", _lib.ptr(x), _lib.ptr(table), _lib.ptr(out), x.numel(), table.numel(), _s())
    return out


def sum_rows_f32(x, rows, cols, row_stride=None, offset=0):
    """out[c] = sum_r x.flat[offset + r * row_stride + c] (fp32): the batch reduction of an embedding gradient."""
    _lib.require_gpu(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("sum_rows_f32 takes a contiguous fp32 tensor")
    row_stride = cols if row_stride is None else row_stride
    if offset + (rows - 1) * row_stride + cols > x.numel():
        raise ValueError("sum_rows_f32: window exceeds the tensor")
    out = torch.empty(cols, dtype=torch.float32, device=x.device)
    _lib.call("chb_sum_rows_f32", ctypes.c_void_p(x.data_ptr() + 4 * offset), int(row_stride), int(rows), int(cols), _lib.ptr(out), _s())
    return out


# ------------------------------------------------------------------ general attention (masks, causal, cross-attention)
def _mask_u8(m, b, t, device):
    if m is None:
        return None
    m = torch.as_tensor(m, device=device)
    if tuple(m.shape) != (b, t):
        raise ValueError("attention mask must have shape [batch, length] = %s, got %s" % ((b, t), tuple(m.shape)))
    return (m != 0).to(torch.uint8).contiguous()


def attention_general_fwd(q, k, v, b, tq, tk, h, hd, value_mask=None, query_mask=None, causal=False, drop_rate=0.0, drop_key=0, scale=0.0):
    """q [B*Tq, H*hd], k / v [B*Tk, H*hd] bf16 -> (o bf16 [B*Tq, H*hd], lse fp32 [B*H*Tq]); masks uint8 [B, T] or None."""
    _lib.require_gpu(q, k, v, value_mask, query_mask)
    o = torch.empty((b * tq, h * hd), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty(b * h * tq, dtype=torch.float32, device=q.device)
    _lib.call("chb_attention_general_fwd", _lib.ptr(q), q.stride(0), _lib.ptr(k), k.stride(0), _lib.ptr(v), v.stride(0), _lib.ptr(o), o.stride(0),
              _lib.ptr(lse), int(b), int(tq), int(tk), int(h), int(hd), _lib.ptr(value_mask), _lib.ptr(query_mask), int(bool(causal)),
              float(drop_rate), ctypes.c_uint32(int(drop_key)), float(scale), _s())
    return o, lse


def attention_general_bwd(q, k, v, o, d_o, lse, b, tq, tk, h, hd, value_mask=None, query_mask=None, causal=False, drop_rate=0.0, drop_key=0, scale=0.0):
    """-> (dq fp32 [B*Tq, H*hd], dk, dv fp32 [B*Tk, H*hd])."""
    _lib.require_gpu(q, k, v, o, d_o, lse, value_mask, query_mask)
    dq = torch.empty((b * tq, h * hd), dtype=torch.float32, device=q.device)
    dk = torch.zeros((b * tk, h * hd), dtype=torch.float32, device=q.device)
    dv = torch.zeros((b * tk, h * hd), dtype=torch.float32, device=q.device)
    _lib.call("chb_attention_general_bwd", _lib.ptr(q), q.stride(0), _lib.ptr(k), k.stride(0), _lib.ptr(v), v.stride(0), _lib.ptr(o), o.stride(0),
              _lib.ptr(d_o), d_o.stride(0), _lib.ptr(lse), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), int(b), int(tq), int(tk), int(h), int(hd),
              _lib.ptr(value_mask), _lib.ptr(query_mask), int(bool(causal)), float(drop_rate), ctypes.c_uint32(int(drop_key)), float(scale), _s())
    return dq, dk, dv
