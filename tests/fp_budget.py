"""The floating-point BUDGET model the tolerance of every whole-model comparison against the plain fp32 oracle is derived from
(test infrastructure).  A tolerance here is not "what was measured x 1.5": it is

    budget = SLACK * FLOOR * sqrt(k + kappa^2)

  FLOOR = 1.65e-3: the relative L2 error of ONE bf16 storage rounding of a tensor (8 significant bits; measured on every bf16-stored
          tensor of tests/fp_bar.py between 1.5e-3 and 1.8e-3);
  k     = how many INDEPENDENT bf16 storage roundings lie on the path from the inputs to the tensor (they add like a random walk);
          counted from the build's rounding points:
            forward, per encoder block (6): LayerNorm-1 output, q/k/v, the probabilities P, the attention output o,
              LayerNorm-2 output, gelu(a1)                                       -> logits after L blocks: 6 L
            attention output of block l: its own 4 (h1, qkv, P, o) + the 6 l of the blocks in front
            backward, per block (6): dz of the MLP branch, d(a1), d(h2), dz of the attention branch, dO, dqkv; + dlogits and the
              head's d(embedding) (2); + the forward state the gradient is taken at (6 L)
              -> a gradient formed in block l of L: 2 + 6 (L - 1 - l) + 4 + 6 L     (block 0 of 2: 24; block 0 of 12: 144)
  kappa = the CANCELLATION ratio of a product whose left operand is rounded element by element (dS, P in the attention backward):
          rounding A to bf16 perturbs (A B)_ij by FLOOR * sqrt(sum_k A_ik^2 B_kj^2) in RMS, so relative to ||A B|| the rounding of A
          alone costs FLOOR * kappa, kappa = sqrt(sum (A^2)(B^2)) / ||A B||  (>= 1; large when signed terms cancel - dQ = dS K sums
          197..577 signed dS entries per element).  kappa is COMPUTED from the oracle's own fp32 tensors, per case.
  SLACK = 1.5 covers that the roundings are not exactly independent nor of exactly equal weight.

The column against the bf16-EMULATING oracle (same rounding points as the build) is the kernel-defect detector and keeps tight,
measured bounds: both sides share the storage roundings, what is left are rounding flips and accumulation order."""
import math

import torch

FLOOR = 1.65e-3
SLACK = 1.5


def k_forward(depth):
    return 6 * depth


def k_attention_output(block):
    return 4 + 6 * block


def k_backward(depth, block):
    return 2 + 6 * (depth - 1 - block) + 4 + k_forward(depth)


def budget(k, kappa=0.0):
    return SLACK * FLOOR * math.sqrt(k + kappa * kappa)


def cancellation(a, b):
    """kappa of the product a @ b when a is rounded element-wise (batched over leading axes)."""
    a, b = a.double(), b.double()
    num = torch.sqrt(torch.matmul(a * a, b * b).sum())
    return float(num / (torch.matmul(a, b).norm() + 1e-300))


def attention_kappas(q, k, v, o, do, keep=None, inv_keep=1.0):
    """Cancellation ratios of the attention backward (csrc/attention.hip); q, k, v, o, do: [B, H, N, hd] fp32 oracle tensors.
      dq, dk: dS rounded to bf16 element-wise in front of dS.K / dS^T.Q;  dv: the dropped probabilities in front of P^T.dO;
      dq also: delta_i = sum_d dO_id O_id is formed from the STORED bf16 O and dO (two roundings per term).  An error e_i of
      delta_i shifts the whole row dS_i. by -P_i. e_i - coherently over the keys - so dQ_i moves by -e_i (P K)_i, while the same
      errors enter dK_j = sum_i dS_ij Q_i with independent signs.  RMS e_i = FLOOR * sqrt(2 sum_d (dO_id O_id)^2), hence
      kappa_delta = sqrt(sum_i 2 sum_d (dO_id O_id)^2 ||(P K)_i||^2) / ||dS K||.
    (Measured on MI355X at the BASELINE geometries: kappa of the dS rounding is 0.9-1.3 - there is NO cancellation blow-up in dS.K,
    an error-compensated dS would buy nothing; kappa_delta is what sets dQ apart from dK / dV.)"""
    scale = 1.0 / math.sqrt(q.shape[-1])
    q, k, v, o, do = (t.double() for t in (q, k, v, o, do))
    p = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * scale, dim=-1)
    keepc = keep.to(p.dtype) * inv_keep if keep is not None else 1.0
    delta = (do * o).sum(dim=-1, keepdim=True)
    dp = torch.matmul(do, v.transpose(-1, -2))
    ds = p * (dp * keepc - delta)
    pd = p * keepc
    dq = torch.matmul(ds, k)
    e2 = 2.0 * ((do * o) ** 2).sum(dim=-1)                        # [B, H, N]: variance of delta_i in units of FLOOR^2
    pk2 = (torch.matmul(p, k) ** 2).sum(dim=-1)                    # ||(P K)_i||^2
    kappa_delta = float(torch.sqrt((e2 * pk2).sum()) / (dq.norm() + 1e-300))
    k_ds = cancellation(ds, k)
    return {"dq": math.sqrt(k_ds * k_ds + kappa_delta * kappa_delta), "dq_rounding_of_dS": k_ds, "dq_delta": kappa_delta,
            "dk": cancellation(ds.transpose(-1, -2), q), "dv": cancellation(pd.transpose(-1, -2), do)}


def budget_for_row(row, depth, kappa=None):
    """Budget against the plain fp32 oracle for a row of tests/fp_bar.py at `depth` blocks."""
    if row == "logits" or row == "loss per sample":      # a per-sample loss is a 1-Lipschitz function of the centred logits
        return budget(k_forward(depth))
    if row == "o (block 0)":
        return budget(k_attention_output(0))
    if row == "o (last block)":
        return budget(k_attention_output(depth - 1))
    if row.split(" ")[0] in ("dq", "dk", "dv"):
        return budget(k_backward(depth, 0), kappa or 0.0)
    return budget(k_backward(depth, 0))                   # dO (block 0) and every weight-gradient family (the worst path: block 0)
