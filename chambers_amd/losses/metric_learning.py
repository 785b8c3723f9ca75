"""chambers.losses.metric_learning on MI355X (reference: chambers/losses/metric_learning.py): MultiSimilarityLoss (:124-178),
MultiSimilarityLossMatrix (:181-235), ContrastiveLoss (:238-287), NTXentLoss (:290-323).

Same constructor arguments and get_config() keys.  `loss(y_true, y_pred)` returns the Keras loss value (mean over the batch of
the per-anchor losses); `value_and_gradient` also returns d(loss)/d(y_pred), which feeds `ViTEngine.backward(doutput)` (through
`L2Normalization.backward` when the embeddings are normalised).  One fused HIP launch pair (chambers_amd/csrc/metric.hip):
pairwise similarities, positive / negative masks, the miner's thresholds, the log-sum-exp terms and the pair weights never leave
the device."""
import torch

from .. import _lib
from ..miners import MultiSimilarityMiner


def _stream():
    return torch.cuda.current_stream().cuda_stream


class MultiSimilarityLoss:
    def __init__(self, pos_scale=2.0, neg_scale=40.0, threshold=0.5, ignore_diag=True, ignore_negative_labels=True,
                 miner="default", name="multi_similarity_loss", **kwargs):
        self.pos_scale, self.neg_scale, self.threshold = float(pos_scale), float(neg_scale), float(threshold)
        self.ignore_diag, self.ignore_negative_labels = bool(ignore_diag), bool(ignore_negative_labels)
        self.miner = MultiSimilarityMiner(margin=0.1) if isinstance(miner, str) and miner == "default" else miner   # :150
        if self.miner is not None and not isinstance(self.miner, MultiSimilarityMiner):
            raise NotImplementedError("only MultiSimilarityMiner (or None) is fused into the loss kernel")
        self.name = name

    def _run(self, y_true, y_pred, want_grad):
        _lib.require_gpu(y_pred)
        if y_pred.dim() != 2:
            raise ValueError("y_pred must be [n, embedding dim], got %s" % (tuple(y_pred.shape),))
        emb = y_pred.to(torch.float32).contiguous()
        b, d = emb.shape
        labels = torch.as_tensor(y_true, device=emb.device).reshape(-1).to(torch.int32).contiguous()
        if labels.numel() != b:
            raise ValueError("y_true must hold one label per embedding")
        rows = torch.empty(b, dtype=torch.float32, device=emb.device)
        grad = torch.empty_like(emb) if want_grad else None
        ws = torch.empty((b, b), dtype=torch.float32, device=emb.device) if want_grad else None
        _lib.call("chb_multi_similarity_loss", _lib.ptr(emb), _lib.ptr(labels), _lib.ptr(rows), _lib.ptr(ws), _lib.ptr(grad), b, d,
                  self.pos_scale, self.neg_scale, self.threshold, float(self.miner.margin) if self.miner is not None else 0.0,
                  1 if self.miner is not None else 0, int(self.ignore_diag), int(self.ignore_negative_labels), _stream())
        return rows, grad

    def per_sample(self, y_true, y_pred):
        """`call` of the reference (losses/metric_learning.py:29-51): the per-anchor loss vector."""
        return self._run(y_true, y_pred, False)[0]

    def __call__(self, y_true, y_pred):
        return self.per_sample(y_true, y_pred).mean()          # keras Loss reduction: SUM_OVER_BATCH_SIZE

    loss = __call__

    def value_and_gradient(self, y_true, y_pred):
        rows, grad = self._run(y_true, y_pred, True)
        return rows.mean(), grad

    def get_config(self):
        return {"name": self.name, "pos_scale": self.pos_scale, "neg_scale": self.neg_scale, "threshold": self.threshold,
                "ignore_diag": self.ignore_diag, "ignore_negative_labels": self.ignore_negative_labels,
                "miner": None if self.miner is None else self.miner.get_config()}


def _miner_args(miner):
    if miner is not None and not isinstance(miner, MultiSimilarityMiner):
        raise NotImplementedError("only MultiSimilarityMiner (or None) is fused into the loss kernels")
    return (float(miner.margin) if miner is not None else 0.0), (1 if miner is not None else 0)


def _emb_labels(y_true, y_pred):
    _lib.require_gpu(y_pred)
    if y_pred.dim() != 2:
        raise ValueError("y_pred must be [n, embedding dim], got %s" % (tuple(y_pred.shape),))
    emb = y_pred.to(torch.float32).contiguous()
    labels = torch.as_tensor(y_true, device=emb.device).reshape(-1).to(torch.int32).contiguous()
    if labels.numel() != emb.shape[0]:
        raise ValueError("y_true must hold one label per embedding")
    return emb, labels


class _RowLoss:
    """per_sample / __call__ / value_and_gradient over a `_run(y_true, y_pred, want_grad) -> (rows, grad)`."""

    def per_sample(self, y_true, y_pred):
        return self._run(y_true, y_pred, False)[0]

    def __call__(self, y_true, y_pred):
        return self.per_sample(y_true, y_pred).mean()          # keras Loss reduction: SUM_OVER_BATCH_SIZE

    loss = __call__

    def value_and_gradient(self, y_true, y_pred):
        rows, grad = self._run(y_true, y_pred, True)
        return rows.mean(), grad


class MultiSimilarityLossMatrix(_RowLoss):
    """:181-235 (PairMatrixLoss :112-121): y_pred is the [n, n] similarity matrix itself, y_true its positive-pair mask (cast to
    bool); the gradient is taken with respect to the matrix."""

    def __init__(self, pos_scale=2.0, neg_scale=40.0, threshold=0.5, ignore_diag=True, ignore_negative_labels=True,
                 miner="default", name="multi_similarity_loss", **kwargs):
        self.pos_scale, self.neg_scale, self.threshold = float(pos_scale), float(neg_scale), float(threshold)
        self.ignore_diag, self.ignore_negative_labels = bool(ignore_diag), bool(ignore_negative_labels)
        self.miner = MultiSimilarityMiner(margin=0.1) if isinstance(miner, str) and miner == "default" else miner
        _miner_args(self.miner)
        self.name = name

    def _run(self, y_true, y_pred, want_grad):
        _lib.require_gpu(y_pred)
        if y_pred.dim() != 2 or y_pred.shape[0] != y_pred.shape[1]:
            raise ValueError("y_pred must be the square similarity matrix, got %s" % (tuple(y_pred.shape),))
        sim = y_pred.to(torch.float32).contiguous()
        b = sim.shape[0]
        mask = (torch.as_tensor(y_true, device=sim.device) != 0).to(torch.uint8).contiguous()
        if tuple(mask.shape) != (b, b):
            raise ValueError("y_true must be the [n, n] positive-pair mask")
        rows = torch.empty(b, dtype=torch.float32, device=sim.device)
        grad = torch.empty_like(sim) if want_grad else None
        margin, use = _miner_args(self.miner)
        _lib.call("chb_multi_similarity_loss_matrix", _lib.ptr(sim), _lib.ptr(mask), _lib.ptr(rows), _lib.ptr(grad), b, self.pos_scale,
                  self.neg_scale, self.threshold, margin, use, int(self.ignore_diag), _stream())
        return rows, grad

    def get_config(self):
        return {"name": self.name, "pos_scale": self.pos_scale, "neg_scale": self.neg_scale, "threshold": self.threshold,
                "ignore_diag": self.ignore_diag, "ignore_negative_labels": self.ignore_negative_labels,
                "miner": None if self.miner is None else self.miner.get_config()}


class ContrastiveLoss(_RowLoss):
    """:238-287: positives pay (positive_margin - s)^e / e, negatives max(0, s - negative_margin)^e / e."""

    def __init__(self, positive_margin=1.0, negative_margin=0.3, exponent=2, ignore_diag=True, ignore_negative_labels=True,
                 miner=None, name="contrastive_loss", **kwargs):
        self.positive_margin, self.negative_margin, self.exponent = positive_margin, negative_margin, exponent
        self.ignore_diag, self.ignore_negative_labels = bool(ignore_diag), bool(ignore_negative_labels)
        self.miner = miner
        _miner_args(self.miner)
        self.name = name

    def _run(self, y_true, y_pred, want_grad):
        emb, labels = _emb_labels(y_true, y_pred)
        b, d = emb.shape
        rows = torch.empty(b, dtype=torch.float32, device=emb.device)
        grad = torch.empty_like(emb) if want_grad else None
        ws = torch.empty((b, b), dtype=torch.float32, device=emb.device) if want_grad else None
        margin, use = _miner_args(self.miner)
        _lib.call("chb_contrastive_loss", _lib.ptr(emb), _lib.ptr(labels), _lib.ptr(rows), _lib.ptr(ws), _lib.ptr(grad), b, d,
                  float(self.positive_margin), float(self.negative_margin), float(self.exponent), margin, use, int(self.ignore_diag),
                  int(self.ignore_negative_labels), _stream())
        return rows, grad

    def get_config(self):
        return {"name": self.name, "positive_margin": self.positive_margin, "negative_margin": self.negative_margin, "exponent": self.exponent,
                "ignore_diag": self.ignore_diag, "ignore_negative_labels": self.ignore_negative_labels,
                "miner": None if self.miner is None else self.miner.get_config()}


class NTXentLoss(_RowLoss):
    """:290-323: categorical cross-entropy of the temperature-scaled similarity rows (diagonal at -1e9) against the multi-hot rows of
    same-label pairs; from_logits=False (the reference's default) is Keras' probability form: rows divided by their sum and clipped
    to [1e-7, 1 - 1e-7] before the logarithm."""

    def __init__(self, temperature=1.0, from_logits=False, name=None, **kwargs):
        self.temperature, self.from_logits = temperature, bool(from_logits)
        self.name = name

    def _run(self, y_true, y_pred, want_grad):
        emb, labels = _emb_labels(y_true, y_pred)
        b, d = emb.shape
        rows = torch.empty(b, dtype=torch.float32, device=emb.device)
        grad = torch.empty_like(emb) if want_grad else None
        ws = torch.empty((b, b), dtype=torch.float32, device=emb.device) if want_grad else None
        _lib.call("chb_ntxent_loss", _lib.ptr(emb), _lib.ptr(labels), _lib.ptr(rows), _lib.ptr(ws), _lib.ptr(grad), b, d, float(self.temperature),
                  int(self.from_logits), _stream())
        return rows, grad

    def get_config(self):
        return {"name": self.name, "temperature": self.temperature, "from_logits": self.from_logits}
