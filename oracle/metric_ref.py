"""CPU oracle of the metric-learning head — TEST INFRASTRUCTURE ONLY (torch-CPU, autograd for gradients).

Restates chambers/losses/metric_learning.py:29-323 (PairLoss.call, get_signed_pairs, the compute_loss of MultiSimilarityLoss,
MultiSimilarityLossMatrix and ContrastiveLoss, NTXentLoss.call with Keras' categorical_crossentropy [UPSTREAM-RECALLED]),
chambers/miners.py:48-60 (MultiSimilarityMiner.compute_masks) and chambers/layers/normalization.py:16-18 (tf.nn.l2_normalize:
x * rsqrt(max(sum x^2, 1e-12))) on dense masks instead of ragged tensors: an empty ragged row reduces to the dtype's lowest /
highest value under tf.reduce_max / reduce_min, which the masked max / min below reproduce.  Parity unpinned (no reference
test exercises the loss)."""
import torch


def l2_normalize(x):
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim=-1, keepdim=True), min=1e-12))


def multi_similarity_loss(y_true, y_pred, pos_scale=2.0, neg_scale=40.0, threshold=0.5, ignore_diag=True,
                          ignore_negative_labels=True, miner_margin=0.1):
    """Returns the per-anchor loss vector (the Keras loss value is its mean).  miner_margin=None -> miner=None."""
    y = y_true.reshape(-1, 1)
    sim = y_pred @ y_pred.t()
    pos = y == y.t()
    neg = ~pos
    if ignore_negative_labels:
        keep = y >= 0                      # broadcasts over columns: anchors with a negative label have no pairs (:87-90)
        pos, neg = pos & keep, neg & keep
    if ignore_diag:
        eye = torch.eye(sim.shape[0], dtype=torch.bool)
        pos, neg = pos & ~eye, neg & ~eye
    if miner_margin is not None:
        big = torch.finfo(sim.dtype).max
        pos_thresh = torch.where(neg, sim, torch.full_like(sim, -big)).max(dim=1).values + miner_margin
        neg_thresh = torch.where(pos, sim, torch.full_like(sim, big)).min(dim=1).values - miner_margin
        pos = pos & (sim < pos_thresh.reshape(-1, 1))
        neg = neg & (sim > neg_thresh.reshape(-1, 1))
    zero = torch.zeros_like(sim)
    sp = torch.where(pos, torch.exp(-pos_scale * (sim - threshold)), zero).sum(dim=1)
    sn = torch.where(neg, torch.exp(neg_scale * (sim - threshold)), zero).sum(dim=1)
    return torch.log(1 + sp) / pos_scale + torch.log(1 + sn) / neg_scale


def _pair_masks(y_true, sim, matrix, ignore_diag, ignore_negative_labels, miner_margin):
    if matrix:                               # PairMatrixLoss.compute_signed_masks (:117-121): y_true cast to bool
        pos = y_true != 0
        neg = ~pos
        keep = torch.ones_like(pos)          # (:87-90 on a boolean matrix: never negative)
    else:
        y = y_true.reshape(-1, 1)
        pos = y == y.t()
        neg = ~pos
        keep = (y >= 0).expand_as(pos)
    if ignore_negative_labels:
        pos, neg = pos & keep, neg & keep
    if ignore_diag:
        eye = torch.eye(sim.shape[0], dtype=torch.bool)
        pos, neg = pos & ~eye, neg & ~eye
    if miner_margin is not None:
        big = torch.finfo(sim.dtype).max
        pos_thresh = torch.where(neg, sim, torch.full_like(sim, -big)).max(dim=1).values + miner_margin
        neg_thresh = torch.where(pos, sim, torch.full_like(sim, big)).min(dim=1).values - miner_margin
        pos = pos & (sim < pos_thresh.reshape(-1, 1))
        neg = neg & (sim > neg_thresh.reshape(-1, 1))
    return pos, neg


def multi_similarity_loss_matrix(y_true, y_pred, pos_scale=2.0, neg_scale=40.0, threshold=0.5, ignore_diag=True, miner_margin=0.1):
    """:181-235: y_pred is the similarity matrix, y_true its positive mask."""
    pos, neg = _pair_masks(y_true, y_pred, True, ignore_diag, True, miner_margin)
    zero = torch.zeros_like(y_pred)
    sp = torch.where(pos, torch.exp(-pos_scale * (y_pred - threshold)), zero).sum(dim=1)
    sn = torch.where(neg, torch.exp(neg_scale * (y_pred - threshold)), zero).sum(dim=1)
    return torch.log(1 + sp) / pos_scale + torch.log(1 + sn) / neg_scale


def contrastive_loss(y_true, y_pred, positive_margin=1.0, negative_margin=0.3, exponent=2, ignore_diag=True, ignore_negative_labels=True,
                     miner_margin=None):
    """:238-287 (per-anchor vector)."""
    sim = y_pred @ y_pred.t()
    pos, neg = _pair_masks(y_true, sim, False, ignore_diag, ignore_negative_labels, miner_margin)
    zero = torch.zeros_like(sim)
    pl = torch.where(pos, torch.pow(positive_margin - sim, exponent) / exponent, zero).sum(dim=1)
    nl = torch.where(neg, torch.pow(torch.clamp(sim - negative_margin, min=0), exponent) / exponent, zero).sum(dim=1)
    return pl + nl


def ntxent_loss(y_true, y_pred, temperature=1.0, from_logits=False):
    """:290-323 (per-anchor vector; the Keras value is its mean).  tf.keras.losses.CategoricalCrossentropy [UPSTREAM-RECALLED]:
    from_logits -> softmax_cross_entropy_with_logits(labels, logits) = -sum_j y_j log softmax(z)_j (labels are NOT normalised);
    otherwise output / sum(output), clip to [1e-7, 1 - 1e-7], -sum_j y_j log(output_j)."""
    n = y_pred.shape[0]
    sim = (y_pred @ y_pred.t()) / temperature
    eye = torch.eye(n, dtype=torch.bool)
    sim = torch.where(eye, torch.full_like(sim, -1e9), sim)
    y = y_true.reshape(-1, 1)
    onehot = ((y == y.t()) & ~eye).to(sim.dtype)
    if from_logits:
        return -(onehot * torch.log_softmax(sim, dim=1)).sum(dim=1)
    out = sim / sim.sum(dim=1, keepdim=True)
    out = torch.clamp(out, 1e-7, 1 - 1e-7)
    return -(onehot * torch.log(out)).sum(dim=1)
