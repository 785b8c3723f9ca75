"""CPU oracle of the metric-learning head — TEST INFRASTRUCTURE ONLY (torch-CPU, autograd for gradients).

Restates chambers/losses/metric_learning.py:29-178 (PairLoss.call, get_signed_pairs, MultiSimilarityLoss.compute_loss),
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
