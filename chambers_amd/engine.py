"""Whole-model ViT executor for MI355X: forward, backward, AdamW and data-parallel gradient
exchange over preallocated HBM buffers, every arithmetic step a hand-written HIP kernel behind
the C ABI (chambers_amd/kernels.py).  torch only allocates, zero-fills and provides streams /
torch.distributed (RCCL).

What it replaces in the reference: the Keras graph built by VisionTransformer
(models/backbones/vision_transformer.py:235-283) executed by `Model.fit` with
chambers.optimizers.AdamW (optimizers.py:372-464) — i.e. SURVEY §3.2-3.4.

Numerics ("mixed_bfloat16"-like, utils/generic.py:32-40): fp32 master weights, bf16 GEMM/attention
operands, fp32 accumulation; the residual stream and all gradients of it are kept in fp32 (one
step more precise than Keras' bf16 activations).  Dropout masks come from the counter hash
(chambers_amd/rng.py), regenerated in backward.

Parameter memory: ONE flat fp32 buffer (+ same-shaped grad / Adam m / Adam v), laid out in the
order gradients become final during backward (head, final norm, block L-1 ... block 0,
embeddings) so that data-parallel buckets are contiguous slices that can be all-reduced while
earlier blocks are still in backward.  Two bf16 images of the matrices are refreshed after every
optimizer step: [K][N] (B operand of dgrad) and its transpose [N][K] (B operand of forward).
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib
from . import kernels as K
from . import rng

ALIGN = 1024  # parameter tensors start at multiples of 1024 elements (16-byte vectors, AdamW decay chunks)


def _round_up(x, m):
    return (x + m - 1) // m * m


class ViTConfig:
    def __init__(self, patch_size, patch_dim, n_encoder_layers, n_heads, ff_dim, dropout_rate=0.1, image_size=(224, 224),
                 classes=1000, include_top=True, feature_dim=None, pooling="cls", norm_epsilon=1e-6, norm_mode="tf",
                 distilled=False, return_dist_token=True):
        self.patch_size, self.patch_dim, self.n_encoder_layers = int(patch_size), int(patch_dim), int(n_encoder_layers)
        self.n_heads, self.ff_dim, self.dropout_rate = int(n_heads), int(ff_dim), float(dropout_rate)
        self.image_size = (int(image_size[0]), int(image_size[1]))
        self.classes, self.include_top, self.feature_dim = int(classes), bool(include_top), feature_dim
        self.pooling, self.norm_epsilon, self.norm_mode = pooling, float(norm_epsilon), norm_mode
        if self.patch_dim % self.n_heads:
            raise ValueError("patch_dim must be divisible by n_heads")
        self.head_dim = self.patch_dim // self.n_heads
        # GEMM-friendly widths of the two heads (zero-padded columns / rows; the pad never leaves the engine)
        self.classes_pad = _round_up(self.classes, 64)
        self.feature_pad = _round_up(int(feature_dim), 64) if feature_dim else 0
        self.grid = (self.image_size[0] // self.patch_size, self.image_size[1] // self.patch_size)
        self.n_patches = self.grid[0] * self.grid[1]
        # DistilledVisionTransformer (vision_transformer.py:295-400): a distillation token after the class token, a second head
        self.distilled, self.return_dist_token = bool(distilled), bool(return_dist_token)
        if self.distilled and feature_dim:
            raise ValueError("the distilled variant has no feature head (vision_transformer.py:295-311)")
        self.n_special = 2 if self.distilled else 1
        self.n_tokens = self.n_patches + self.n_special
        self.patch_k = self.patch_size * self.patch_size * 3

    def as_oracle_cfg(self):
        return {"patch_size": self.patch_size, "n_encoder_layers": self.n_encoder_layers, "n_heads": self.n_heads,
                "dropout_rate": self.dropout_rate, "norm_epsilon": self.norm_epsilon, "pooling": None if self.pooling == "none" else self.pooling,
                "return_dist_token": self.return_dist_token}


class ParamSpec:
    __slots__ = ("name", "shape", "offset", "size", "matrix", "decay")

    def __init__(self, name, shape, offset, matrix, decay):
        self.name, self.shape, self.offset, self.matrix, self.decay = name, tuple(shape), offset, matrix, decay
        self.size = int(np.prod(shape))


def build_param_table(cfg, decay_fn=None):
    """Internal parameter list in backward-completion order, with 1024-aligned offsets.
    Returns (specs, total_elements, bucket_ranges) — bucket k is the slice whose gradients are final after stage k of
    backward: 0 = heads + final norm; for the s-th block from the top (s = 1..L) 2s-1 = its MLP, norm2 and projection
    (final before its attention backward) and 2s = its QKV and norm1; 2L+1 = embeddings."""
    d, ff, n, kp = cfg.patch_dim, cfg.ff_dim, cfg.n_tokens, cfg.patch_k
    specs, buckets = [], []
    off = 0

    def add(name, shape, matrix):
        nonlocal off
        decay = True if decay_fn is None else bool(decay_fn(name))
        s = ParamSpec(name, shape, off, matrix, decay)
        specs.append(s)
        off = _round_up(off + s.size, ALIGN)

    start = off
    if cfg.include_top:
        cpad = cfg.classes_pad
        in_dim = cfg.feature_pad or d
        add("predictions/kernel", (in_dim, cpad), True)
        add("predictions/bias", (cpad,), False)
        if cfg.distilled:
            add("predictions_dist/kernel", (d, cpad), True)
            add("predictions_dist/bias", (cpad,), False)
    if cfg.feature_dim:
        add("feature/kernel", (d, cfg.feature_pad), True)
        add("feature/bias", (cfg.feature_pad,), False)
    add("encoder/norm/gamma", (d,), False)
    add("encoder/norm/beta", (d,), False)
    buckets.append((start, off))
    for i in reversed(range(cfg.n_encoder_layers)):
        start = off
        p = "encoder/layer_%d/" % i
        add(p + "dense2/kernel", (ff, d), True)
        add(p + "dense2/bias", (d,), False)
        add(p + "dense1/kernel", (d, ff), True)
        add(p + "dense1/bias", (ff,), False)
        add(p + "norm2/gamma", (d,), False)
        add(p + "norm2/beta", (d,), False)
        add(p + "proj/kernel", (d, d), True)
        add(p + "proj/bias", (d,), False)
        buckets.append((start, off))
        start = off
        add(p + "qkv/kernel", (d, 3 * d), True)
        add(p + "qkv/bias", (3 * d,), False)
        add(p + "norm1/gamma", (d,), False)
        add(p + "norm1/beta", (d,), False)
        buckets.append((start, off))
    start = off
    add("pos_embedding/embeddings", (n, d), False)
    add("add_cls_token/embeddings", (cfg.n_special, d), False)     # row 0 class token, row 1 distillation token (distilled variant)
    add("patch_embeddings/embedding/kernel", (kp, d), True)
    add("patch_embeddings/embedding/bias", (d,), False)
    buckets.append((start, off))
    return specs, off, buckets


def keras_variable_names(cfg):
    """Internal tensor name -> the `var.name` strings of the Keras variables it holds, as the reference model would
    name them (scope = nesting of layer names; un-named sublayers get Keras' per-class auto names in construction
    order: EncoderLayer.__init__ builds multi_head_attention, norm1, dense1, dense2, norm2 — layers/transformer.py:31-49 —
    and Encoder builds its layers, then the output norm, :271-283).  [UPSTREAM-RECALLED: Keras auto-naming.]
    The optimizer's decay_include / decay_exclude regexes are matched against these (optimizers.py:169-181)."""
    def auto(base, k):
        return base if k == 0 else "%s_%d" % (base, k)

    names = {"patch_embeddings/embedding/kernel": ["patch_embeddings/embedding/kernel:0"],
             "patch_embeddings/embedding/bias": ["patch_embeddings/embedding/bias:0"],
             "add_cls_token/embeddings": ["add_cls_token/embeddings:0"] + (["add_dist_token/embeddings:0"] if cfg.distilled else []),
             "pos_embedding/embeddings": ["pos_embedding/embeddings:0"]}
    L = cfg.n_encoder_layers
    for i in range(L):
        p = "encoder/layer_%d/" % i
        scope = "encoder/%s/" % auto("encoder_layer", i)
        mha = scope + auto("multi_head_attention", i) + "/"
        names[p + "qkv/kernel"] = [mha + "w_query:0", mha + "w_key:0", mha + "w_value:0"]
        names[p + "qkv/bias"] = [mha + "b_query:0", mha + "b_key:0", mha + "b_value:0"]
        names[p + "proj/kernel"], names[p + "proj/bias"] = [mha + "w_projection:0"], [mha + "b_projection:0"]
        for j, nm in enumerate(("norm1", "norm2")):
            ln = scope + auto("layer_normalization", 2 * i + j) + "/"
            names[p + nm + "/gamma"], names[p + nm + "/beta"] = [ln + "gamma:0"], [ln + "beta:0"]
        for j, nm in enumerate(("dense1", "dense2")):
            dn = scope + auto("dense", 2 * i + j) + "/"
            names[p + nm + "/kernel"], names[p + nm + "/bias"] = [dn + "kernel:0"], [dn + "bias:0"]
    ln = "encoder/" + auto("layer_normalization", 2 * L) + "/"
    names["encoder/norm/gamma"], names["encoder/norm/beta"] = [ln + "gamma:0"], [ln + "beta:0"]
    if cfg.feature_dim:
        names["feature/kernel"], names["feature/bias"] = ["feature/kernel:0"], ["feature/bias:0"]
    if cfg.include_top:
        names["predictions/kernel"], names["predictions/bias"] = ["predictions/kernel:0"], ["predictions/bias:0"]
        if cfg.distilled:
            names["predictions_dist/kernel"], names["predictions_dist/bias"] = ["predictions_dist/kernel:0"], ["predictions_dist/bias:0"]
    return names


def decay_fn_from_variable_predicate(cfg, allowed):
    """Lift a per-Keras-variable predicate (name -> bool) to the engine's tensors.  The fused QKV kernel / bias hold
    three Keras variables; the decay flag is per tensor, so a predicate that separates them is refused."""
    table = keras_variable_names(cfg)

    def fn(internal_name):
        votes = {bool(allowed(v)) for v in table[internal_name]}
        if len(votes) != 1:
            raise ValueError("weight-decay filter separates %s, which are stored as one fused tensor (%s)"
                             % (", ".join(table[internal_name]), internal_name))
        return votes.pop()
    return fn


# ---------------------------------------------------------------------------------------------
# Keras-layout <-> internal-layout conversion (SURVEY §8b "weight naming / ownership")
# ---------------------------------------------------------------------------------------------
def keras_to_internal(kw, cfg):
    """kw: dict of Keras-named numpy arrays (names as in oracle/vit_ref.py).  Returns internal dict."""
    d, h, hd = cfg.patch_dim, cfg.n_heads, cfg.head_dim
    out = {}
    out["patch_embeddings/embedding/kernel"] = np.asarray(kw["patch_embeddings/embedding/kernel"]).reshape(cfg.patch_k, d)
    out["patch_embeddings/embedding/bias"] = np.asarray(kw["patch_embeddings/embedding/bias"])
    toks = [np.asarray(kw["add_cls_token/embeddings"]).reshape(1, d)]
    if cfg.distilled:
        toks.append(np.asarray(kw["add_dist_token/embeddings"]).reshape(1, d))
    out["add_cls_token/embeddings"] = np.concatenate(toks, axis=0)
    out["pos_embedding/embeddings"] = np.asarray(kw["pos_embedding/embeddings"])
    for i in range(cfg.n_encoder_layers):
        p = "encoder/layer_%d/" % i
        a = p + "multi_head_attention/"
        wq, wk, wv = (np.asarray(kw[a + k]).reshape(d, h * hd) for k in ("w_query", "w_key", "w_value"))
        bq, bk, bv = (np.asarray(kw[a + k]).reshape(h * hd) for k in ("b_query", "b_key", "b_value"))
        out[p + "qkv/kernel"] = np.concatenate([wq, wk, wv], axis=1)
        out[p + "qkv/bias"] = np.concatenate([bq, bk, bv])
        out[p + "proj/kernel"] = np.asarray(kw[a + "w_projection"]).transpose(0, 2, 1).reshape(h * hd, d)
        out[p + "proj/bias"] = np.asarray(kw[a + "b_projection"]).reshape(d)
        for k in ("norm1/gamma", "norm1/beta", "norm2/gamma", "norm2/beta", "dense1/kernel", "dense1/bias", "dense2/kernel", "dense2/bias"):
            out[p + k] = np.asarray(kw[p + k])
    out["encoder/norm/gamma"] = np.asarray(kw["encoder/norm/gamma"])
    out["encoder/norm/beta"] = np.asarray(kw["encoder/norm/beta"])
    if cfg.feature_dim:
        fk = np.zeros((d, cfg.feature_pad), dtype=np.float32)
        fk[:, :cfg.feature_dim] = np.asarray(kw["feature/kernel"])
        fb = np.zeros((cfg.feature_pad,), dtype=np.float32)
        fb[:cfg.feature_dim] = np.asarray(kw["feature/bias"])
        out["feature/kernel"], out["feature/bias"] = fk, fb
    if cfg.include_top:
        cpad = cfg.classes_pad
        k = np.asarray(kw["predictions/kernel"])
        kp = np.zeros((cfg.feature_pad or d, cpad), dtype=np.float32)
        kp[:k.shape[0], :cfg.classes] = k
        bp = np.zeros((cpad,), dtype=np.float32)
        bp[:cfg.classes] = np.asarray(kw["predictions/bias"])
        out["predictions/kernel"], out["predictions/bias"] = kp, bp
        if cfg.distilled:
            kd = np.zeros((d, cpad), dtype=np.float32)
            kd[:, :cfg.classes] = np.asarray(kw["predictions_dist/kernel"])
            bd = np.zeros((cpad,), dtype=np.float32)
            bd[:cfg.classes] = np.asarray(kw["predictions_dist/bias"])
            out["predictions_dist/kernel"], out["predictions_dist/bias"] = kd, bd
    return out


def internal_to_keras(iw, cfg):
    d, h, hd = cfg.patch_dim, cfg.n_heads, cfg.head_dim
    p_ = cfg.patch_size
    out = {}
    out["patch_embeddings/embedding/kernel"] = iw["patch_embeddings/embedding/kernel"].reshape(p_, p_, 3, d)
    out["patch_embeddings/embedding/bias"] = iw["patch_embeddings/embedding/bias"]
    out["add_cls_token/embeddings"] = iw["add_cls_token/embeddings"].reshape(cfg.n_special, d)[0:1]
    if cfg.distilled:
        out["add_dist_token/embeddings"] = iw["add_cls_token/embeddings"].reshape(cfg.n_special, d)[1:2]
    out["pos_embedding/embeddings"] = iw["pos_embedding/embeddings"]
    for i in range(cfg.n_encoder_layers):
        p = "encoder/layer_%d/" % i
        a = p + "multi_head_attention/"
        w = iw[p + "qkv/kernel"]
        b = iw[p + "qkv/bias"]
        for j, nm in enumerate(("query", "key", "value")):
            out[a + "w_" + nm] = w[:, j * d:(j + 1) * d].reshape(d, h, hd)
            out[a + "b_" + nm] = b[j * d:(j + 1) * d].reshape(h, 1, hd)
        out[a + "w_projection"] = iw[p + "proj/kernel"].reshape(h, hd, d).transpose(0, 2, 1)
        out[a + "b_projection"] = iw[p + "proj/bias"].reshape(1, d)
        for k in ("norm1/gamma", "norm1/beta", "norm2/gamma", "norm2/beta", "dense1/kernel", "dense1/bias", "dense2/kernel", "dense2/bias"):
            out[p + k] = iw[p + k]
    out["encoder/norm/gamma"] = iw["encoder/norm/gamma"]
    out["encoder/norm/beta"] = iw["encoder/norm/beta"]
    if cfg.feature_dim:
        out["feature/kernel"], out["feature/bias"] = iw["feature/kernel"][:, :cfg.feature_dim], iw["feature/bias"][:cfg.feature_dim]
    if cfg.include_top:
        out["predictions/kernel"] = iw["predictions/kernel"][:cfg.feature_dim or d, :cfg.classes]
        out["predictions/bias"] = iw["predictions/bias"][:cfg.classes]
        if cfg.distilled:
            out["predictions_dist/kernel"] = iw["predictions_dist/kernel"][:, :cfg.classes]
            out["predictions_dist/bias"] = iw["predictions_dist/bias"][:cfg.classes]
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def init_keras_weights(cfg, seed=1234):
    """Random initial weights exactly as the reference's initialisers prescribe (glorot-uniform dense /
    attention kernels, zero biases, TruncatedNormal(0.02) cls / pos, LayerNorm ones/zeros)."""
    from . import initializers as I
    I.set_seed(seed)
    d, h, hd, ff, p = cfg.patch_dim, cfg.n_heads, cfg.head_dim, cfg.ff_dim, cfg.patch_size
    tn = I.TruncatedNormal(stddev=0.02)
    kw = {"patch_embeddings/embedding/kernel": I.glorot_uniform((p, p, 3, d)), "patch_embeddings/embedding/bias": I.zeros((d,)),
          "add_cls_token/embeddings": tn((1, d)), "pos_embedding/embeddings": tn((cfg.n_tokens, d))}
    for i in range(cfg.n_encoder_layers):
        pre = "encoder/layer_%d/" % i
        a = pre + "multi_head_attention/"
        kw[a + "w_query"], kw[a + "b_query"] = I.glorot_uniform((d, h, hd)), I.zeros((h, 1, hd))
        kw[a + "w_value"], kw[a + "b_value"] = I.glorot_uniform((d, h, hd)), I.zeros((h, 1, hd))
        kw[a + "w_key"], kw[a + "b_key"] = I.glorot_uniform((d, h, hd)), I.zeros((h, 1, hd))
        kw[a + "w_projection"], kw[a + "b_projection"] = I.glorot_uniform((h, d, hd)), I.zeros((1, d))
        kw[pre + "norm1/gamma"], kw[pre + "norm1/beta"] = I.ones((d,)), I.zeros((d,))
        kw[pre + "dense1/kernel"], kw[pre + "dense1/bias"] = I.glorot_uniform((d, ff)), I.zeros((ff,))
        kw[pre + "dense2/kernel"], kw[pre + "dense2/bias"] = I.glorot_uniform((ff, d)), I.zeros((d,))
        kw[pre + "norm2/gamma"], kw[pre + "norm2/beta"] = I.ones((d,)), I.zeros((d,))
    kw["encoder/norm/gamma"], kw["encoder/norm/beta"] = I.ones((d,)), I.zeros((d,))
    if cfg.feature_dim:
        kw["feature/kernel"], kw["feature/bias"] = I.glorot_uniform((d, cfg.feature_dim)), I.zeros((cfg.feature_dim,))
    if cfg.distilled:
        kw["add_dist_token/embeddings"] = tn((1, d))
    if cfg.include_top:
        kw["predictions/kernel"] = I.glorot_uniform((cfg.feature_dim or d, cfg.classes))
        kw["predictions/bias"] = I.zeros((cfg.classes,))
        if cfg.distilled:
            kw["predictions_dist/kernel"], kw["predictions_dist/bias"] = I.glorot_uniform((d, cfg.classes)), I.zeros((cfg.classes,))
    return kw


# ---------------------------------------------------------------------------------------------
# data-parallel gradient exchange: device-agnostic so gloo/CPU tests cover it
# ---------------------------------------------------------------------------------------------
class GradBucketReducer:
    """All-reduces contiguous slices of a flat gradient buffer once backward marks them final.
    One logical all-reduce (sum) per step, issued as a handful of asynchronous collectives so the
    exchange of block l overlaps the backward of blocks < l; `finish()` waits for all of them.
    Averaging (x 1/world) is folded into the optimizer's grad_scale.

    `bucket_ready(k)` only queues the bucket; `flush()` issues what is queued.  The engine flushes right before a kernel with a
    large grid of short workgroups (attention backward): the collective's workgroups then take their CUs from a launch the
    hardware dispatcher load-balances, not from a persistent GEMM whose late workgroups would stretch that launch by the
    collective's whole duration (measured with a stand-in collective, tools/rccl_contention.py)."""

    def __init__(self, flat_grad, buckets, process_group=None, payload="fp32", force=False):
        """payload: "fp32" (default: the gradient slices are reduced in place) or "bf16" (each slice is rounded to bf16 into a
        staging buffer, reduced there, and widened back into the flat fp32 buffer by finish(): half the bytes on xGMI - 173 MB instead
        of 346 MB per step for ViT-B/16 - at one bf16 rounding of every rank's contribution plus the collective's bf16 sums).
        force: stay active with a process group of ONE rank - every collective is then really issued (init, async handles, stream
        ordering, finish()) and returns its input: the RCCL path exercised on a single GPU (bench.py --force-dp)."""
        import torch.distributed as dist
        if payload not in ("fp32", "bf16"):
            raise ValueError("payload must be 'fp32' or 'bf16', got %r" % (payload,))
        self.dist = dist
        self.flat = flat_grad
        self.buckets = list(buckets)
        self.group = process_group
        ready = dist.is_available() and dist.is_initialized()
        self.active = ready and (dist.get_world_size(process_group) > 1 or bool(force))
        self.world = dist.get_world_size(process_group) if self.active else 1
        self.payload = payload
        self.handles = []
        self.queued = []
        self.staged = []          # bf16 payload: (lo, hi, staging buffer) of the collectives in flight
        # accounting for bench.py's data-parallel report: collectives issued, bytes all-reduced, and (measure=True) HIP-event
        # pairs around finish()'s waits - the time the compute stream stood still for the exchange
        self.n_collectives = 0
        self.bytes_reduced = 0
        self.measure = False
        self.exposed_events = []

    def bucket_ready(self, k):
        if self.active:
            self.queued.append(k)

    def _narrow(self, lo, hi):
        """fp32 slice -> a fresh bf16 staging buffer (HIP cast on the GPU; the CPU form exists for the gloo tests only)."""
        src = self.flat[lo:hi]
        if src.is_cuda:
            from . import kernels as K
            return K.cast_bf16(src)
        return src.to(torch.bfloat16)

    def _widen(self, lo, hi, buf):
        if buf.is_cuda:
            from . import _lib, kernels as K
            _lib.call("chb_cast_bf16_f32", _lib.ptr(buf), _lib.ptr(self.flat[lo:hi]), hi - lo, K._s())
        else:
            self.flat[lo:hi].copy_(buf)

    def flush(self):
        """One collective per run of adjacent queued buckets (they are contiguous slices of the flat buffer)."""
        ranges = []
        for k in self.queued:
            lo, hi = self.buckets[k]
            if hi <= lo:
                continue
            if ranges and ranges[-1][1] == lo:
                ranges[-1][1] = hi
            else:
                ranges.append([lo, hi])
        self.queued = []
        for lo, hi in ranges:
            if self.payload == "bf16":
                buf = self._narrow(lo, hi)
                self.staged.append((lo, hi, buf))
                self.handles.append(self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.bytes_reduced += (hi - lo) * 2
            else:
                self.handles.append(self.dist.all_reduce(self.flat[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.bytes_reduced += (hi - lo) * self.flat.element_size()
            self.n_collectives += 1

    def finish(self):
        self.flush()
        if not self.handles:
            return
        timed = self.measure and self.flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in self.handles:
            h.wait()
        if timed:
            e1.record()
            self.exposed_events.append((e0, e1))
        for lo, hi, buf in self.staged:      # behind the waits: the compute stream is ordered after every collective
            self._widen(lo, hi, buf)
        self.staged = []
        self.handles = []

    def exposed_ms(self):
        """Sum over the recorded finish() calls of the compute stream's wait for the exchange (call after a device synchronise)."""
        return float(sum(a.elapsed_time(b) for a, b in self.exposed_events))

    @property
    def grad_scale(self):
        return 1.0 / self.world


class _SideRing:
    """Same-shaped buffers written on the main stream and read by launches on a side stream: `acquire()` hands out the next slot
    for writing once the side stream's last read of it is done; `mark_read(stream)` notes a read just enqueued there."""

    def __init__(self, bufs):
        self.bufs = list(bufs)
        self.i = 0
        self.read_ev = [None] * len(self.bufs)

    @property
    def cur(self):
        return self.bufs[self.i]

    def acquire(self):
        self.i = (self.i + 1) % len(self.bufs)
        ev = self.read_ev[self.i]
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self.read_ev[self.i] = None
        return self.bufs[self.i]

    def mark_read(self, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        self.read_ev[self.i] = ev


# ---------------------------------------------------------------------------------------------
class ViTEngine:
    def __init__(self, cfg, batch_size, device=None, training=True, seed=0, decay_fn=None, process_group=None, overlap_wgrad=None,
                 pad_m=None, grad_payload=None, force_dp=False):
        if not torch.cuda.is_available():
            raise RuntimeError("ViTEngine needs an MI355X (torch.cuda is not available); there is no CPU fallback")
        self.cfg, self.B, self.training, self.seed = cfg, int(batch_size), bool(training), int(seed)
        self.dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if cfg.head_dim != 64:
            raise ValueError("attention kernels are built for head_dim 64 (all reference ViT configs), got %d" % cfg.head_dim)
        if cfg.pooling not in ("cls", "avg", "max", "sum", "none"):
            raise ValueError("pooling must be one of 'cls', 'avg', 'max', 'sum', None (vision_transformer.py:172-191); got %r" % (cfg.pooling,))
        if cfg.pooling == "none" and (cfg.include_top or cfg.feature_dim):
            raise ValueError("pooling=None returns the token sequence; heads on the token axis are not built (use a pooling mode)")
        self.specs, self.n_params_padded, self.buckets = build_param_table(cfg, decay_fn)
        self.by_name = {s.name: s for s in self.specs}
        dev, f32, bf = self.dev, torch.float32, torch.bfloat16
        nflat = self.n_params_padded
        self.P = torch.zeros(nflat, dtype=f32, device=dev)
        self.Pb = torch.zeros(nflat, dtype=bf, device=dev)    # [K][N] images (dgrad B operand)
        self.Pbt = torch.zeros(nflat, dtype=bf, device=dev)   # [N][K] images (forward B operand)
        if training:
            self.G = torch.zeros(nflat, dtype=f32, device=dev)
            self.Mo = torch.zeros(nflat, dtype=f32, device=dev)
            self.Vo = torch.zeros(nflat, dtype=f32, device=dev)
            self._upload_decay_flags()
            self.reducer = GradBucketReducer(self.G, self.buckets, process_group,
                                             payload=grad_payload or os.environ.get("CHB_GRAD_PAYLOAD", "fp32"), force=force_dp)
            _lib.call("chb_gemm_tile_queue_reset", K._s())      # the tile-queue counters start a run clean (include/chambers_hip.h)
            # The persistent GEMMs' tile queue (CHB_GEMM_TILE_QUEUE=1: late workgroups draw fewer tiles instead of stretching the
            # launch by a co-running collective's duration; tools/rccl_contention.py) stays OPT-IN: the engine does not
            # switch it on by itself, not even for data-parallel runs, until a real multi-GPU RCCL run has confirmed bit-equal
            # output and clean counters there.
            self._g_clean = True        # G holds zeros (allocation, or the last AdamW launch cleared it behind its read)
        mats = [s for s in self.specs if s.matrix]
        desc = np.array([[s.offset, s.offset, s.shape[0], s.shape[1]] for s in mats], dtype=np.int64)
        self.ct_desc = torch.as_tensor(desc, device=dev)
        self.ct_n = len(mats)
        self.ct_tiles = max(((s.shape[0] + 63) // 64) * ((s.shape[1] + 63) // 64) for s in mats)
        # AdamW (+ the bf16 operand refresh) of a parameter bucket as soon as backward has marked it final, on a stream of its own:
        # the update is HBM-bound and its short workgroups run in the CUs a persistent GEMM's last partial round leaves idle
        # (train_step only - the hyper-parameters must be known before backward - and single-process only: with a process group
        # the exchange comes first).  OPT-IN (CHB_EARLY_ADAMW=1): measured on MI355X it hides nothing - 68.19-68.51 ms per step
        # against 68.11-68.35 with the one launch after backward, same box, alternating runs.
        self.early_adamw = training and os.environ.get("CHB_EARLY_ADAMW", "0") == "1"
        self._early_hp = None
        self.opt_stream = None
        self.ct_bucket = []
        for lo, hi in (self.buckets if training else ()):
            rows = [[s.offset, s.offset, s.shape[0], s.shape[1]] for s in mats if lo <= s.offset < hi]
            self.ct_bucket.append((torch.as_tensor(np.array(rows, dtype=np.int64).reshape(-1, 4), device=dev), len(rows),
                                   max([((r[2] + 63) // 64) * ((r[3] + 63) // 64) for r in rows] + [0])))
        self.opt_step = 0
        self._overlap_arg = overlap_wgrad
        self._pad_arg = pad_m
        self._alloc_activations()

    def _upload_decay_flags(self):
        flags = np.zeros(self.n_params_padded // ALIGN, dtype=np.uint8)
        for s in self.specs:
            flags[s.offset // ALIGN:_round_up(s.offset + s.size, ALIGN) // ALIGN] = 1 if s.decay else 0
        self.decay_flags = torch.as_tensor(flags, device=self.dev)

    def set_decay_fn(self, decay_fn):
        """Re-derive which tensors AdamW decays (optimizers.py:169-181) without touching weights, Adam moments or the step
        count: `Model.compile` with another optimizer keeps the training state, as Keras does."""
        for s in self.specs:
            s.decay = True if decay_fn is None else bool(decay_fn(s.name))
        if self.training:
            self._upload_decay_flags()

    # ---- views --------------------------------------------------------------------------
    def _v(self, buf, name):
        s = self.by_name[name]
        return buf[s.offset:s.offset + s.size].view(*s.shape)

    def p(self, name):
        return self._v(self.P, name)

    def g(self, name):
        return self._v(self.G, name)

    def wb(self, name):   # bf16 [K][N]
        return self._v(self.Pb, name)

    def wbt(self, name):  # bf16 [N][K]
        s = self.by_name[name]
        return self.Pbt[s.offset:s.offset + s.size].view(s.shape[1], s.shape[0])

    # ---- weights in / out ------------------------------------------------------------------
    def load_keras_weights(self, kw):
        iw = keras_to_internal(kw, self.cfg)
        host = np.zeros(self.n_params_padded, dtype=np.float32)
        for s in self.specs:
            arr = np.asarray(iw[s.name], dtype=np.float32)
            if arr.shape != s.shape:
                raise ValueError("weight %s: expected shape %s, got %s" % (s.name, s.shape, arr.shape))
            host[s.offset:s.offset + s.size] = arr.reshape(-1)
        self.P.copy_(torch.from_numpy(host))
        self.refresh_operands()

    def export_keras_weights(self):
        host = self.P.detach().cpu().numpy()
        iw = {s.name: host[s.offset:s.offset + s.size].reshape(s.shape).copy() for s in self.specs}
        return internal_to_keras(iw, self.cfg)

    def export_keras_grads(self):
        """Gradients of the last backward in the Keras layout.  adamw_step(zero_grad=True) - the default, and what train_step /
        Model.train_step / optimizer.apply run - clears the gradient buffer behind its read: asking for gradients after it is an
        error, not a dictionary of zeros (call this between backward() and the optimizer step, or step with zero_grad=False)."""
        if self._g_clean:
            raise RuntimeError("no gradients to export: the buffer was cleared by adamw_step(zero_grad=True) (or no backward has run); "
                               "export between backward() and the optimizer step, or pass zero_grad=False")
        host = self.G.detach().cpu().numpy()
        iw = {s.name: host[s.offset:s.offset + s.size].reshape(s.shape).copy() for s in self.specs}
        return internal_to_keras(iw, self.cfg)

    def refresh_operands(self):
        """fp32 master -> bf16 [K][N] and [N][K] images of every matrix (one launch)."""
        K.cast_transpose(self.P, self.Pb, self.Pbt, self.ct_desc, self.ct_n, self.ct_tiles)

    # ---- buffers --------------------------------------------------------------------------
    def _alloc_activations(self):
        cfg, B, dev = self.cfg, self.B, self.dev
        f32, bf = torch.float32, torch.bfloat16
        d, ff, n = cfg.patch_dim, cfg.ff_dim, cfg.n_tokens
        self.M = B * n
        self.Mp = _round_up(self.M, 256)
        # Rows the block GEMMs (forward projections and dgrad) are LAUNCHED over.  Every token matrix has Mp rows, so a ragged
        # M = batch x tokens (config 5: 128 x 577 = 288.5 tiles) can run as Mp = full 256-row tiles and take the pipelined full-tile
        # kernel instead of the lockstep one with clamped staging and a guarded epilogue (5-15 % slower).  What makes that safe:
        #  * forward pad rows hold junk (bias, gelu(bias), ...: finite) that nothing reads: LayerNorm, attention, pooling, the loss and
        #    every reduction over tokens run over the true M (or over batch elements);
        #  * backward pad rows are exactly ZERO and stay zero: the gradient buffers start as zeros, LayerNorm / attention / dropout
        #    backward write the true M rows only, and a dgrad GEMM maps zero rows to zero rows (no bias in backward; the gelu'-multiply
        #    multiplies 0 by a finite saved value), so the fused column sums and the weight-gradient GEMMs (which already reduce over
        #    Mp rows) add nothing for them.
        # Small problems (below the persistent kernel's threshold) keep the true M.  CHB_PAD_M=0 / pad_m=False: launch over M.
        pad = bool(int(os.environ.get("CHB_PAD_M", "1"))) if self._pad_arg is None else bool(self._pad_arg)
        self.Mg = self.Mp if (pad and self.M >= 2048) else self.M
        self.Bp = _round_up(B, 64)
        self.Mpatch = B * cfg.n_patches
        self.Mpatch_p = _round_up(self.Mpatch, 64)
        Mp = self.Mp
        L = cfg.n_encoder_layers
        nsave = L if self.training else 1
        z = lambda *shape, dtype=bf: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        self.patches = z(self.Mpatch_p, cfg.patch_k)
        self.xs = [z(Mp, d, dtype=f32) for _ in range((L + 1) if self.training else 2)]
        self.acts = []
        # keep bits of the dropout on the attention probabilities, written by the forward and tested by the backward (one word pair
        # per query and lane group: 32 bytes per query row, 39 MB per block at batch 512 / 197 tokens)
        self.use_drop_bits = self.training and cfg.dropout_rate > 0.0 and n <= 224
        for _ in range(nsave):
            self.acts.append({
                "h1": z(Mp, d), "mean1": z(Mp, dtype=f32), "rstd1": z(Mp, dtype=f32), "qkv": z(Mp, 3 * d), "o": z(Mp, d),
                "lse": z(B * cfg.n_heads * n, dtype=f32), "xmid": z(Mp, d, dtype=f32), "h2": z(Mp, d), "mean2": z(Mp, dtype=f32),
                "rstd2": z(Mp, dtype=f32), "a1": z(Mp, ff), "u": z(Mp, ff),
                "drop_bits": K.attention_drop_bits(B, n, cfg.n_heads, device=dev) if self.use_drop_bits else None})
        self.hf = z(self.Bp, d)                      # pooled, normalised embedding (bf16 operand of the heads)
        nstat = self.Bp if cfg.pooling == "cls" else Mp
        self.meanf, self.rstdf = z(nstat, dtype=f32), z(nstat, dtype=f32)
        if cfg.pooling != "cls":
            self.hn = z(Mp, d)                       # final LayerNorm over every token (avg / max / sum pooling)
            self.pool_arg = torch.zeros(self.Bp, d, dtype=torch.int32, device=dev) if cfg.pooling == "max" else None
        F = cfg.feature_pad
        if F:
            self.feat = z(self.Bp, F, dtype=f32)     # tanh(feature) — the model output when include_top=False
            self.feat_b = z(self.Bp, F)
        if cfg.include_top:
            self.cpad = cfg.classes_pad
            self.logits = z(self.Bp, self.cpad, dtype=f32)
        if cfg.distilled:
            self.hfd = z(self.Bp, d)                 # normalised distillation-token embedding
            self.meand, self.rstdd = z(self.Bp, dtype=f32), z(self.Bp, dtype=f32)
            if cfg.include_top:
                self.logits_dist = z(self.Bp, self.cpad, dtype=f32)
        self.loss_vec = z(self.Bp, dtype=f32)
        if self.training:
            if cfg.include_top:
                self.dlogits = z(self.Bp, self.cpad)
                if cfg.distilled:
                    self.dlogits_dist = z(self.Bp, self.cpad)
            if cfg.distilled:
                self.dhfd = z(self.Bp, d)
            if F:
                self.dfeat = z(self.Bp, F, dtype=f32)
                self.dfz = z(self.Bp, F)
            self.dhf = z(self.Bp, d)
            self.dx = z(Mp, d, dtype=f32)
            self.dz = z(Mp, d)
            self.da1 = z(Mp, ff)
            self.dh = z(Mp, d)
            self.do = z(Mp, d)
            self.dqkv = z(Mp, 3 * d)
            # scratch for the split-K partial planes of the weight-gradient GEMMs (one launch at a time on the stream)
            self.tn_ws = torch.empty(max(K.tn_workspace_elems(*sp.shape) for sp in self.specs if sp.matrix), dtype=f32, device=dev)
            # Weight gradients on a side stream (overlap_wgrad): a block's four wgrad GEMMs depend only on saved activations and on
            # dz / da1 / dqkv, and nothing but the optimizer reads their output, so they need not sit in the dgrad chain.  Kernels
            # of ONE stream are separated by full barriers - a launch's tail (4.6 rounds of tiles on the N = 768 GEMMs), the ~6 us
            # between dependent launches and the bandwidth-bound LayerNorm / latency-bound attention backward leave CUs idle that
            # an independent MFMA-bound launch of another stream fills.  The three operands get small rings so the side stream
            # may run up to a block behind; it has its own split-K scratch.
            # Measured (bench.py, same box, A/B): 73.1 -> 71.7 ms/step on two boxes, 75.6 -> 75.0 on a third (+0.8 ... +1.9 %).  Off
            # by default: every launch's duration then includes time it shared the chip, which makes per-kernel timings (the
            # rooflines in profiles/) unreadable - opt in with CHB_OVERLAP_WGRAD=1 or ViTEngine(..., overlap_wgrad=True).
            self.overlap_wgrad = bool(int(os.environ.get("CHB_OVERLAP_WGRAD", "0"))) if self._overlap_arg is None else bool(self._overlap_arg)
            if self.overlap_wgrad:
                self.side = torch.cuda.Stream(device=dev)
                self.tn_ws_side = torch.empty_like(self.tn_ws)
                self.dz_ring = _SideRing([self.dz, z(Mp, d), z(Mp, d), z(Mp, d)])
                self.da1_ring = _SideRing([self.da1, z(Mp, ff)])
                self.dqkv_ring = _SideRing([self.dqkv, z(Mp, 3 * d)])
            self.dpatch = z(self.Mpatch_p, d)
            self.labels = torch.zeros(self.Bp, dtype=torch.int32, device=dev)
        # One C-ABI call per encoder block and direction (chb_vit_block_fwd / _bwd: the same launches in the same order, issued from
        # C): the records are filled once - every buffer is static - and only the dropout keys (and, at inference, the ping-pong
        # x_in / x_out) change per call.  CHB_ENGINE_PY_BLOCKS=1 keeps the launch-by-launch Python path (tests compare the two).
        self.c_blocks = not bool(int(os.environ.get("CHB_ENGINE_PY_BLOCKS", "0")))
        # Where a data-parallel run starts a block's all-reduce (C block path).  "block" (default): behind the block's last launch - the
        # collective's workgroups then share the chip with the next block's GEMMs.  "attention": in the middle of the block, right
        # before the attention backward (the round-3 choice, made for the lean attention kernel with its 6144 short workgroups; the
        # persistent pipelined backward holds every CU for its whole duration, a collective started beside it is pushed behind it onto
        # the weight-gradient GEMM, which has no tile queue: tools/rccl_contention.py measured +3.5 ms a step for it against +0.9 ... +2.4
        # for "block"; profiles/r04_collective_contention.txt).
        self.dp_flush = os.environ.get("CHB_DP_FLUSH", "block")
        if self.dp_flush not in ("block", "attention"):
            raise ValueError("CHB_DP_FLUSH must be 'block' or 'attention', got %r" % (self.dp_flush,))
        self._build_block_records()

    def _build_block_records(self):
        cfg = self.cfg
        # CHB_FOLD_BATCH=1: one split-K scratch per weight gradient of a block, so that ONE launch folds the four
        # (chb_gemm_tn_fold_multi): 16 fold launches a step instead of 49.  Measured and left OFF: 67.22 / 67.24 ms
        # per step against 67.01 / 66.99 with a fold behind every GEMM (same box, alternating runs) - a fold that follows its GEMM at
        # once finds the 66 MB of planes in the Infinity Cache; four of them at the end of the block come from HBM.
        self.tn_ws4 = None
        if self.training and self.c_blocks and bool(int(os.environ.get("CHB_FOLD_BATCH", "0"))):
            self.tn_ws4 = torch.empty(4 * self.tn_ws.numel(), dtype=torch.float32, device=self.dev)
        L, d = cfg.n_encoder_layers, cfg.patch_dim
        dp = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        self.blocks = []
        for l in range(L):
            pre = "encoder/layer_%d/" % l
            a = self.acts[l if self.training else 0]
            r = _lib.VitBlock()
            r.B, r.N, r.H, r.hd, r.D, r.FF = self.B, cfg.n_tokens, cfg.n_heads, cfg.head_dim, d, cfg.ff_dim
            r.M, r.Mg, r.Mp = self.M, self.Mg, self.Mp
            r.eps, r.drop_rate = cfg.norm_epsilon, cfg.dropout_rate
            r.emit_dz = 1 if l > 0 else 0
            for f, n in (("ln1_gamma", "norm1/gamma"), ("ln1_beta", "norm1/beta"), ("ln2_gamma", "norm2/gamma"), ("ln2_beta", "norm2/beta"),
                         ("qkv_bias", "qkv/bias"), ("proj_bias", "proj/bias"), ("fc1_bias", "dense1/bias"), ("fc2_bias", "dense2/bias")):
                setattr(r, f, dp(self.p(pre + n)))
            for f, n in (("qkv", "qkv/kernel"), ("proj", "proj/kernel"), ("fc1", "dense1/kernel"), ("fc2", "dense2/kernel")):
                setattr(r, f + "_wt", dp(self.wbt(pre + n)))
                setattr(r, f + "_w", dp(self.wb(pre + n)))
            for f in ("h1", "qkv", "o", "h2", "a1", "u", "mean1", "rstd1", "lse", "xmid", "mean2", "rstd2", "drop_bits"):
                setattr(r, f, dp(a[f]))
            if self.training:
                r.x_in, r.x_out = dp(self.xs[l]), dp(self.xs[l + 1])
                for f, n in (("g_ln1_gamma", "norm1/gamma"), ("g_ln1_beta", "norm1/beta"), ("g_ln2_gamma", "norm2/gamma"), ("g_ln2_beta", "norm2/beta"),
                             ("g_qkv_bias", "qkv/bias"), ("g_proj_bias", "proj/bias"), ("g_fc1_bias", "dense1/bias"), ("g_qkv_w", "qkv/kernel"),
                             ("g_proj_w", "proj/kernel"), ("g_fc1_w", "dense1/kernel"), ("g_fc2_w", "dense2/kernel")):
                    setattr(r, f, dp(self.g(pre + n)))
                if l > 0:
                    r.g_prev_fc2_bias = dp(self.g("encoder/layer_%d/dense2/bias" % (l - 1)))
                for f in ("dx", "dz", "da1", "dh", "dqkv"):
                    setattr(r, f, dp(getattr(self, f)))
                r.d_o = dp(self.do)
                r.tn_ws, r.tn_ws_bytes = dp(self.tn_ws), self.tn_ws.numel() * 4
                r.tn_ws_side = dp(self.tn_ws_side) if self.overlap_wgrad else None
                r.tn_ws4 = dp(self.tn_ws4) if self.tn_ws4 is not None else None
            self.blocks.append(r)

    def activation_bytes(self):
        tot = 0
        for t in [self.patches, self.hf] + self.xs + [v for a in self.acts for v in a.values() if v is not None]:
            tot += t.numel() * t.element_size()
        return tot

    # ---- forward --------------------------------------------------------------------------
    def _keys(self, training):
        cfg = self.cfg
        rate = cfg.dropout_rate if training else 0.0
        step = self.opt_step
        return rate, (lambda site: rng.site_key(self.seed, step, site))

    def embed(self, images_u8, training, prepatched=False, augment=None):
        """[scheme chain +] normalise + patchify + patch-embedding GEMM (+bias +pos, dropout) + cls row -> xs[0].
        augment: a kernels.AugPlan (RandAugment.plan / AutoAugment.plan) applied to images_u8 inside the patchify pass."""
        cfg = self.cfg
        rate, key = self._keys(training)
        if not prepatched:
            if tuple(images_u8.shape) != (self.B, cfg.image_size[0], cfg.image_size[1], 3):
                raise ValueError("expected images of shape %s, got %s" % ((self.B,) + cfg.image_size + (3,), tuple(images_u8.shape)))
            per_image = isinstance(augment, K.AugItemsPlan)       # an elementwise scheme: every image its own chain
            if augment is not None and len(augment) and cfg.norm_mode == "tf":
                # scheme chain + normalise + patch gather in one pass over the uint8 batch (chb_aug_fused / chb_aug_fused_items)
                (K.aug_fused_items if per_image else K.aug_fused)(images_u8, augment, patch=cfg.patch_size, out=self.patches)
            else:
                if augment is not None and len(augment):
                    images_u8 = (K.aug_fused_items if per_image else K.aug_fused)(images_u8, augment)
                K.normalize_patchify(images_u8, cfg.patch_size, cfg.norm_mode, out=self.patches)
        x0 = self.xs[0]
        K.gemm_nt(self.patches, self.wbt("patch_embeddings/embedding/kernel"), x0, m=self.Mpatch,
                  bias=self.p("patch_embeddings/embedding/bias"), epilogue=K.EPI_PATCH, resid=self.p("pos_embedding/embeddings"),
                  period=cfg.n_patches | ((cfg.n_special - 1) << 24), drop_rate=rate, drop_key=key(rng.SITE_EMBED))
        toks = self.p("add_cls_token/embeddings")
        for row in range(cfg.n_special):     # class token, then the distillation token of the distilled variant
            K.token_row(x0, toks[row], self.p("pos_embedding/embeddings"), self.B, cfg.n_tokens, cfg.patch_dim, row,
                        drop_rate=rate, drop_key=key(rng.SITE_EMBED))
        return x0

    def block_forward(self, l, x_in, x_out, a, training):
        cfg = self.cfg
        rate, key = self._keys(training)
        if self.c_blocks:
            r = self.blocks[l]
            r.key_attn, r.key_proj, r.key_mlp = key(rng.site_attn(l)), key(rng.site_proj(l)), key(rng.site_mlp(l))
            if not self.training:
                r.x_in, r.x_out = x_in.data_ptr(), x_out.data_ptr()
            _lib.call("chb_vit_block_fwd", ctypes.byref(r), 1 if (training and rate) else 0, K._s())
            return
        d, M, Mg = cfg.patch_dim, self.M, self.Mg
        pre = "encoder/layer_%d/" % l
        K.layernorm_fwd(x_in, d, self.p(pre + "norm1/gamma"), self.p(pre + "norm1/beta"), a["h1"], a["mean1"], a["rstd1"], M, d,
                        cfg.norm_epsilon)
        K.gemm_nt(a["h1"], self.wbt(pre + "qkv/kernel"), a["qkv"], m=Mg, bias=self.p(pre + "qkv/bias"))
        K.attention_fwd(a["qkv"], a["o"], a["lse"], self.B, cfg.n_tokens, cfg.n_heads, cfg.head_dim, rate, key(rng.site_attn(l)),
                        drop_bits=a["drop_bits"] if rate else None)
        K.gemm_nt(a["o"], self.wbt(pre + "proj/kernel"), a["xmid"], m=Mg, bias=self.p(pre + "proj/bias"), epilogue=K.EPI_RESID,
                  resid=x_in, drop_rate=rate, drop_key=key(rng.site_proj(l)))
        K.layernorm_fwd(a["xmid"], d, self.p(pre + "norm2/gamma"), self.p(pre + "norm2/beta"), a["h2"], a["mean2"], a["rstd2"], M, d,
                        cfg.norm_epsilon)
        K.gemm_nt(a["h2"], self.wbt(pre + "dense1/kernel"), a["u"], m=Mg, bias=self.p(pre + "dense1/bias"), epilogue=K.EPI_GELU,
                  aux=a["a1"])
        K.gemm_nt(a["u"], self.wbt(pre + "dense2/kernel"), x_out, m=Mg, bias=self.p(pre + "dense2/bias"), epilogue=K.EPI_RESID,
                  resid=a["xmid"], drop_rate=rate, drop_key=key(rng.site_mlp(l)))

    def forward(self, images_u8, training=None, prepatched=False, augment=None):
        """Returns logits fp32 [B, classes] (a view of the padded logits buffer).  prepatched=True: self.patches
        already holds the bf16 patch rows (float32-input path of the Keras-style Model)."""
        training = self.training if training is None else training
        cfg = self.cfg
        L = cfg.n_encoder_layers
        x = self.embed(images_u8, training, prepatched, augment)
        for l in range(L):
            if self.training:
                x_out, a = self.xs[l + 1], self.acts[l]
            else:
                x_out, a = self.xs[(l + 1) & 1], self.acts[0]
            self.block_forward(l, x, x_out, a, training)
            x = x_out
        self.x_final = x
        d, n = cfg.patch_dim, cfg.n_tokens
        if cfg.pooling == "cls":
            # final LayerNorm only where it is consumed: the cls rows (row stride n*d)
            K.layernorm_fwd(x, n * d, self.p("encoder/norm/gamma"), self.p("encoder/norm/beta"), self.hf, self.meanf, self.rstdf, self.B, d,
                            cfg.norm_epsilon)
        else:
            K.layernorm_fwd(x, d, self.p("encoder/norm/gamma"), self.p("encoder/norm/beta"), self.hn, self.meanf, self.rstdf, self.M, d,
                            cfg.norm_epsilon)
            if cfg.pooling == "none":
                return self.hn[:self.M].view(self.B, n, d)
            K.pool_tokens(self.hn, self.hf, self.pool_arg, self.B, n, d, cfg.pooling)
        head_in = self.hf
        if cfg.feature_dim:
            K.gemm_nt(self.hf, self.wbt("feature/kernel"), self.feat, m=self.B, bias=self.p("feature/bias"))
            K.tanh_fwd(self.feat, self.feat_b)
            head_in = self.feat_b
        if cfg.distilled:
            return self._distilled_heads(x, head_in)
        if not cfg.include_top:
            return self.feat[:self.B, :cfg.feature_dim] if cfg.feature_dim else self.hf[:self.B]
        K.gemm_nt(head_in, self.wbt("predictions/kernel"), self.logits, m=self.B, bias=self.p("predictions/bias"))
        return self.logits[:self.B, :cfg.classes]

    def _distilled_heads(self, x, head_in):
        """DistilledVisionTransformer outputs (vision_transformer.py:373-397): the pooled class embedding and the distillation
        token (sequence row 1) each through their own head; a pair, or their average when return_dist_token=False."""
        cfg = self.cfg
        d, n = cfg.patch_dim, cfg.n_tokens
        if cfg.pooling == "cls":
            K.layernorm_fwd(x.view(-1)[d:], n * d, self.p("encoder/norm/gamma"), self.p("encoder/norm/beta"), self.hfd, self.meand, self.rstdd,
                            self.B, d, cfg.norm_epsilon)
        else:
            K.store_rows(self.hfd[:self.B], self.hn[:self.M].view(self.B, n * d)[:, d:2 * d])      # sequence row 1 of every image
        if cfg.include_top:
            K.gemm_nt(head_in, self.wbt("predictions/kernel"), self.logits, m=self.B, bias=self.p("predictions/bias"))
            K.gemm_nt(self.hfd, self.wbt("predictions_dist/kernel"), self.logits_dist, m=self.B, bias=self.p("predictions_dist/bias"))
            if not cfg.return_dist_token:       # the average of the two heads, formed on the padded buffers
                return K.axpby_f32(self.logits, 0.5, self.logits_dist, 0.5)[:self.B, :cfg.classes]
            return self.logits[:self.B, :cfg.classes], self.logits_dist[:self.B, :cfg.classes]
        a, b = K.cast_f32(self.hf[:self.B]), K.cast_f32(self.hfd[:self.B])
        return (a, b) if cfg.return_dist_token else K.axpby_f32(a, 0.5, b, 0.5)

    def capture_inference(self):
        """Capture the inference forward (normalise + patchify -> logits, ~100 launches for ViT-B/16) into a HIP graph over the
        engine's static buffers and return `run(images_u8) -> logits`.  At small batch the eager step is launch-bound (each launch
        is a Python -> ctypes -> hipLaunchKernel round trip); the replay issues the whole chain with one call.  Training steps are
        not captured: their dropout keys are kernel arguments that change every step."""
        cfg = self.cfg
        static_in = torch.zeros((self.B,) + cfg.image_size + (3,), dtype=torch.uint8, device=self.dev)
        self.forward(static_in, training=False)          # warm-up outside the capture (lazy allocations, attribute setup)
        torch.cuda.synchronize(self.dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.forward(static_in, training=False)

        def run(images_u8):
            if tuple(images_u8.shape) != tuple(static_in.shape) or images_u8.dtype != torch.uint8:
                raise ValueError("expected uint8 images of shape %s" % (tuple(static_in.shape),))
            static_in.copy_(images_u8)
            graph.replay()
            return out
        run.graph = graph
        return run

    def loss(self, labels):
        """Sparse softmax cross-entropy from logits, mean over the batch; also fills dlogits when training."""
        self.labels[:self.B].copy_(labels.to(torch.int32))
        K.softmax_ce(self.logits, self.labels, self.loss_vec, self.dlogits if self.training else None, self.cfg.classes, 1.0 / self.B)
        return self.loss_vec[:self.B]

    # ---- backward -------------------------------------------------------------------------
    def backward(self, doutput=None):
        """Backward of the last forward.  With a top, the gradient starts at dlogits (filled by `loss`); a headless model
        (include_top=False: pooled embedding or tanh feature as output) takes d(loss)/d(output) fp32 [B, F] as `doutput`."""
        cfg = self.cfg
        rate, key = self._keys(True)
        d, ff, n, M, Mp, Mg = cfg.patch_dim, cfg.ff_dim, cfg.n_tokens, self.M, self.Mp, self.Mg
        L = cfg.n_encoder_layers
        F = cfg.feature_dim
        # a collective of the previous backward may still be reading / writing slices of G (backward called twice without an
        # optimizer step): drain it before G is touched
        self.reducer.finish()
        if not self._g_clean:
            K.zero_f32(self.G)          # only when backward runs twice without adamw_step; AdamW clears G behind its read
        self._g_clean = False
        # heads
        if cfg.distilled:
            self._distilled_heads_backward(doutput)
        elif cfg.include_top:
            if doutput is not None:
                raise ValueError("doutput is for include_top=False models; with a top the gradient comes from loss()")
            K.gemm_tn(self.feat_b if F else self.hf, self.dlogits, self.g("predictions/kernel"), m=self.Bp, ws=self.tn_ws)
            K.colsum(self.dlogits, self.g("predictions/bias"), m=self.B)
            K.gemm_nt(self.dlogits, self.wb("predictions/kernel"), self.dfeat if F else self.dhf, m=self.B)
        elif cfg.pooling == "none":
            if doutput is None or tuple(doutput.shape) != (self.B, n, d):
                raise ValueError("pooling=None: pass doutput of shape %s" % ((self.B, n, d),))
            K.store_rows(self.dh[:M], doutput.reshape(M, d))
        else:
            if doutput is None or tuple(doutput.shape) != (self.B, F or d):
                raise ValueError("include_top=False: pass doutput of shape %s" % ((self.B, F or d),))
            K.store_rows(self.dfeat[:self.B, :F] if F else self.dhf[:self.B], doutput)
        if F:
            K.tanh_bwd(self.dfeat, self.feat, self.dfz)
            K.gemm_tn(self.hf, self.dfz, self.g("feature/kernel"), m=self.Bp, ws=self.tn_ws)
            K.colsum(self.dfz, self.g("feature/bias"), m=self.B)
            K.gemm_nt(self.dfz, self.wb("feature/kernel"), self.dhf, m=self.B)
        side = self.side if self.overlap_wgrad else None

        def nxt(name):          # next buffer of an operand the side stream reads (a ring slot, or the one buffer)
            if side is None or self.c_blocks:       # the C block entries order re-use of the one buffer by events of their own
                return getattr(self, name)
            buf = getattr(self, name + "_ring").acquire()
            setattr(self, name, buf)      # the attribute always names the latest version (tests and tools read it after backward)
            return buf

        def wgrad(x, name, gname, colsum=None):
            dy = getattr(self, name)
            if side is None:
                K.gemm_tn(x, dy, self.g(gname), m=Mp, ws=self.tn_ws, colsum=colsum)
                return
            ev = torch.cuda.Event()
            ev.record()                   # dy is final on the main stream
            side.wait_event(ev)
            with torch.cuda.stream(side):
                K.gemm_tn(x, dy, self.g(gname), m=Mp, ws=self.tn_ws_side, colsum=colsum)
            getattr(self, name + "_ring").mark_read(side)

        def join():               # gradients written on the side stream are final for whatever the main stream does next
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)

        # pooling + final norm.  The final LayerNorm backward also emits dz of the last block's MLP branch (dropout-backward of the
        # residual gradient, bf16) and its column sums (dense2's bias gradient) - no separate pass over dx.
        last = "encoder/layer_%d/dense2/bias" % (L - 1)
        fuse_tail = not cfg.distilled
        tail = dict(dz=nxt("dz"), dz_colsum=self.g(last), drop_rate=rate, drop_key=key(rng.site_mlp(L - 1))) if fuse_tail else {}
        if cfg.pooling == "cls":
            # the class rows get their gradient, every other row of dx its zero, in one launch (zero_gaps)
            K.layernorm_bwd(self.dhf, self.x_final, n * d, self.meanf, self.rstdf, self.p("encoder/norm/gamma"), self.dx, n * d, False,
                            self.g("encoder/norm/gamma"), self.g("encoder/norm/beta"), self.B, d, zero_gaps=True, **tail)
            if cfg.distilled:   # the distillation token's rows (sequence row 1) of the same LayerNorm
                K.layernorm_bwd(self.dhfd, self.x_final.view(-1)[d:], n * d, self.meand, self.rstdd, self.p("encoder/norm/gamma"),
                                self.dx.view(-1)[d:], n * d, False, self.g("encoder/norm/gamma"), self.g("encoder/norm/beta"), self.B, d)
        else:
            if cfg.pooling != "none":
                K.pool_tokens_bwd(self.dhf, self.pool_arg, self.dh, self.B, n, d, cfg.pooling)
            if cfg.distilled:   # add the distillation head's gradient to row 1 of the normalised sequence
                K.add_rows_bf16(self.dh[:M].view(self.B, n * d)[:, d:2 * d], self.dhfd[:self.B])
            K.layernorm_bwd(self.dh, self.x_final, d, self.meanf, self.rstdf, self.p("encoder/norm/gamma"), self.dx, d, False,
                            self.g("encoder/norm/gamma"), self.g("encoder/norm/beta"), M, d, **tail)
        self._bucket_ready(0)
        if not fuse_tail:     # distilled variant: two LayerNorm launches write dx (class rows, distillation rows); dz follows them
            K.dropout_bwd(self.dx, nxt("dz"), M, d, rate, key(rng.site_mlp(L - 1)))
            K.colsum(self.dz, self.g(last), m=M)
        for l in reversed(range(L) if not self.c_blocks else ()):
            a = self.acts[l]
            pre = "encoder/layer_%d/" % l
            # MLP branch (self.dz = dropout-backward of dx at site_mlp(l))
            wgrad(a["u"], "dz", pre + "dense2/kernel")
            dz = self.dz
            K.gemm_nt(dz, self.wb(pre + "dense2/kernel"), nxt("da1"), m=Mg, epilogue=K.EPI_DGELU, aux=a["a1"],
                      colsum=self.g(pre + "dense1/bias"))           # bias gradient of dense1 fused into the epilogue
            wgrad(a["h2"], "da1", pre + "dense1/kernel")
            K.gemm_nt(self.da1, self.wb(pre + "dense1/kernel"), self.dh, m=Mg)
            K.layernorm_bwd(self.dh, a["xmid"], d, a["mean2"], a["rstd2"], self.p(pre + "norm2/gamma"), self.dx, d, True,
                            self.g(pre + "norm2/gamma"), self.g(pre + "norm2/beta"), M, d, dz=nxt("dz"),
                            dz_colsum=self.g(pre + "proj/bias"), drop_rate=rate, drop_key=key(rng.site_proj(l)))
            # attention branch (self.dz = dropout-backward of dx at site_proj(l))
            wgrad(a["o"], "dz", pre + "proj/kernel")
            K.gemm_nt(self.dz, self.wb(pre + "proj/kernel"), self.do, m=Mg)
            # this block's MLP / projection gradients and the previous block's QKV gradients are final and adjacent in the flat
            # buffer: one all-reduce, started beside the attention backward
            self._bucket_ready(2 * (L - l) - 1)
            if self.reducer.active:
                join()            # the collective reads gradients the side stream wrote
            self.reducer.flush()
            K.attention_bwd(a["qkv"], a["o"], self.do, a["lse"], nxt("dqkv"), self.B, n, cfg.n_heads, cfg.head_dim, rate,
                            key(rng.site_attn(l)), drop_bits=a["drop_bits"] if rate else None)
            # the QKV bias gradient (column sums of dqkv) rides along in the weight-gradient GEMM as ones^T . dqkv: +5 % on that
            # launch instead of a 0.09 ms pass over dqkv (attention_bwd can also fuse it via dbias=, but its four extra
            # accumulators and the cross-wave fold make the 128-register kernel spill: 0.99 ms against 0.69)
            wgrad(a["h1"], "dqkv", pre + "qkv/kernel", colsum=self.g(pre + "qkv/bias"))
            K.gemm_nt(self.dqkv, self.wb(pre + "qkv/kernel"), self.dh, m=Mg)
            if l > 0:
                K.layernorm_bwd(self.dh, self.xs[l], d, a["mean1"], a["rstd1"], self.p(pre + "norm1/gamma"), self.dx, d, True,
                                self.g(pre + "norm1/gamma"), self.g(pre + "norm1/beta"), M, d, dz=nxt("dz"),
                                dz_colsum=self.g("encoder/layer_%d/dense2/bias" % (l - 1)), drop_rate=rate,
                                drop_key=key(rng.site_mlp(l - 1)))
            else:
                K.layernorm_bwd(self.dh, self.xs[l], d, a["mean1"], a["rstd1"], self.p(pre + "norm1/gamma"), self.dx, d, True,
                                self.g(pre + "norm1/gamma"), self.g(pre + "norm1/beta"), M, d)
            self._bucket_ready(2 * (L - l))
        if self.c_blocks:
            main_s = K._s()
            side_p = ctypes.c_void_p(side.cuda_stream) if side is not None else None

            def cjoin():
                if side is not None:
                    _lib.call("chb_side_stream_join", main_s, side_p)

            for l in reversed(range(L)):
                r = self.blocks[l]
                r.key_attn, r.key_proj, r.key_mlp = key(rng.site_attn(l)), key(rng.site_proj(l)), key(rng.site_mlp(l))
                r.key_prev_mlp = key(rng.site_mlp(l - 1)) if l > 0 else 0
                if self.reducer.active and self.dp_flush == "attention":
                    # this block's MLP / projection gradients and the previous block's QKV gradients are final after phase 1 and
                    # adjacent in the flat buffer: one all-reduce, started beside the attention backward
                    _lib.call("chb_vit_block_bwd", ctypes.byref(r), 1, main_s, side_p)
                    self._bucket_ready(2 * (L - l) - 1)
                    cjoin()           # the collective reads gradients the side stream wrote
                    self.reducer.flush()
                    _lib.call("chb_vit_block_bwd", ctypes.byref(r), 2, main_s, side_p)
                    self._bucket_ready(2 * (L - l))
                else:
                    _lib.call("chb_vit_block_bwd", ctypes.byref(r), 3, main_s, side_p)
                    self._bucket_ready(2 * (L - l) - 1)
                    self._bucket_ready(2 * (L - l))
                    if self.reducer.active:       # dp_flush == "block": the block's whole gradient slice as one collective, started here
                        cjoin()
                        self.reducer.flush()
            cjoin()
        join()                    # AdamW (and the last collective) read every gradient
        # embedding stage
        K.embed_bwd(self.dx, self.dpatch, self.g("pos_embedding/embeddings"), self.g("add_cls_token/embeddings"), self.B, n, d, rate,
                    key(rng.SITE_EMBED), n_special=cfg.n_special)
        K.gemm_tn(self.patches, self.dpatch, self.g("patch_embeddings/embedding/kernel"), m=self.Mpatch_p, ws=self.tn_ws,
                  colsum=self.g("patch_embeddings/embedding/bias"))
        self._bucket_ready(2 * L + 1)
        self.reducer.flush()

    def _distilled_heads_backward(self, doutput):
        """d(loss)/d(outputs) of the distilled model: a pair (d_cls, d_dist) of fp32 [B, classes] (or [B, D] without a top), or
        one tensor for return_dist_token=False (the average passes half of it to each head)."""
        cfg = self.cfg
        d = cfg.patch_dim
        if doutput is None:
            raise ValueError("the distilled variant has no built-in loss (the reference defines none): pass doutput")
        if cfg.return_dist_token:
            da, db = doutput
        else:
            da = db = K.axpby_f32(doutput.to(torch.float32) if doutput.dtype != torch.float32 else doutput, 0.5)
        width = cfg.classes if cfg.include_top else d
        if tuple(da.shape) != (self.B, width) or tuple(db.shape) != (self.B, width):
            raise ValueError("doutput tensors must have shape %s" % ((self.B, width),))
        if cfg.include_top:
            K.store_rows(self.dlogits[:self.B, :cfg.classes], da)
            K.store_rows(self.dlogits_dist[:self.B, :cfg.classes], db)
            K.gemm_tn(self.hf, self.dlogits, self.g("predictions/kernel"), m=self.Bp, ws=self.tn_ws)
            K.colsum(self.dlogits, self.g("predictions/bias"), m=self.B)
            K.gemm_nt(self.dlogits, self.wb("predictions/kernel"), self.dhf, m=self.B)
            K.gemm_tn(self.hfd, self.dlogits_dist, self.g("predictions_dist/kernel"), m=self.Bp, ws=self.tn_ws)
            K.colsum(self.dlogits_dist, self.g("predictions_dist/bias"), m=self.B)
            K.gemm_nt(self.dlogits_dist, self.wb("predictions_dist/kernel"), self.dhfd, m=self.B)
        else:
            K.store_rows(self.dhf[:self.B], da)
            K.store_rows(self.dhfd[:self.B], db)

    # ---- optimizer ------------------------------------------------------------------------
    def _bucket_ready(self, k):
        """Backward has enqueued every kernel that writes the gradients of bucket k or reads its weights."""
        self.reducer.bucket_ready(k)
        if self._early_hp is None:
            return
        lo, hi = self.buckets[k]
        if hi <= lo:
            return
        lr_t, b1, b2, eps, wd = self._early_hp
        ev = torch.cuda.Event()
        ev.record()
        self.opt_stream.wait_event(ev)
        with torch.cuda.stream(self.opt_stream):
            K.adamw(self.P[lo:hi], self.G[lo:hi], self.Mo[lo:hi], self.Vo[lo:hi], self.decay_flags[lo // ALIGN:hi // ALIGN], lr_t, b1, b2, eps, wd,
                    1.0, zero_grad=True)
            desc, n_mats, tiles = self.ct_bucket[k]
            if n_mats:
                K.cast_transpose(self.P, self.Pb, self.Pbt, desc, n_mats, tiles)

    @staticmethod
    def _lr_t(learning_rate, beta_1, beta_2, t):
        b1, b2 = np.float32(beta_1), np.float32(beta_2)
        return float(np.float32(learning_rate) * np.sqrt(np.float32(1.0) - np.power(b2, np.float32(t))) / (np.float32(1.0) - np.power(b1, np.float32(t))))

    def adamw_step(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, weight_decay=0.0, zero_grad=True):
        """chambers.optimizers.AdamW semantics (decay first, wd not scaled by lr, keras Adam epsilon-hat form).  zero_grad: the
        update clears each gradient element right after reading it, so the next backward starts from zeros without a fill pass
        (export_keras_grads() must be called before this, or pass zero_grad=False)."""
        self.reducer.finish()
        self.opt_step += 1
        lr_t = self._lr_t(learning_rate, beta_1, beta_2, self.opt_step)
        K.adamw(self.P, self.G, self.Mo, self.Vo, self.decay_flags, float(lr_t), beta_1, beta_2, epsilon, weight_decay,
                self.reducer.grad_scale, zero_grad=zero_grad)
        self._g_clean = bool(zero_grad)
        self.refresh_operands()

    def train_step(self, images_u8, labels, augment=None, **opt):
        """uint8 batch (already augmented, or raw with the scheme's `augment` plan) -> loss vector; runs forward, loss, backward,
        gradient exchange, AdamW."""
        if not self.cfg.include_top or self.cfg.distilled:
            raise ValueError("train_step needs a single classification top (include_top=True, not distilled); drive headless and "
                             "distilled models with forward() + backward(doutput) + adamw_step()")
        self.forward(images_u8, training=True, augment=augment)
        loss = self.loss(labels)
        if self.early_adamw and not self.reducer.active and not self.overlap_wgrad and opt.get("zero_grad", True):
            # per-bucket updates from inside backward (see __init__); same arithmetic as adamw_step, element by element
            if self.opt_stream is None:
                self.opt_stream = torch.cuda.Stream(device=self.dev)
            # (opt_step itself moves only after backward: the dropout keys of this step's backward derive from it, _keys)
            self._early_hp = (self._lr_t(opt.get("learning_rate", 1e-3), opt.get("beta_1", 0.9), opt.get("beta_2", 0.999), self.opt_step + 1),
                              opt.get("beta_1", 0.9), opt.get("beta_2", 0.999), opt.get("epsilon", 1e-7), opt.get("weight_decay", 0.0))
            try:
                self.backward()
            finally:
                self._early_hp = None
            self.opt_step += 1
            torch.cuda.current_stream().wait_stream(self.opt_stream)     # the next forward reads the refreshed operands
            self._g_clean = True
            return loss
        self.backward()
        self.adamw_step(**opt)
        return loss
