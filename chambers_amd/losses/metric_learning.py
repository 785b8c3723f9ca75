"""chambers.losses.metric_learning.MultiSimilarityLoss on MI355X (reference: chambers/losses/metric_learning.py:9-178).

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
