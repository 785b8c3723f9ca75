"""CPU oracle for the chambers ViT hot path (TEST INFRASTRUCTURE ONLY — see
oracle/augment_ref.py header for who may import this).

torch-CPU fp32 restatement, written from the reference text, of
  models/backbones/vision_transformer.py:172-292  (model graph, cls pooling, heads)
  layers/transformer.py:8-77,256-314              (pre-norm EncoderLayer, Encoder)
  layers/attention.py:7-127                       (MultiHeadAttention, ScaledAttention)
  layers/embedding.py:156-182,218-261             (LearnedEmbedding1D, ConcatEmbedding)
  activations.py:46-56                            (exact-erf GELU)
  optimizers.py:147-181,372-464                   (AdamW: decay first, then Adam)

PARITY STATUS: **parity unpinned.**  The reference's only numerical statement for
this path is the manual script test_units/manual_test_vit_weights.py:245-341
(chambers ViT == timm ViT under the weight mapping, atol 1e-5..1e-3), which needs
tensorflow, timm and network access; none can run here.  The arithmetic itself
lives in keras/tensorflow 2.6.0 (LayerNormalization, Dense, Conv2D, Attention,
Dropout, Adam) whose published algorithms are restated below.  `timm_block`
restates the timm block the manual script compares against, and
tests/test_oracle_vit.py checks this oracle equals it under the script's weight
mapping (:27-76), which is the one equivalence the reference does state.
Independent anchors (not the reference): tests/test_oracle_independent.py holds this file against torch's own GELU / LayerNorm /
attention / TransformerEncoderLayer / AdamW and against HuggingFace transformers' ViT and DeiT models (whole forward, random weights).

Gradients come from torch autograd over this forward.  Dropout masks are explicit
(oracle/rng_ref.py defines them).  ``emulate_bf16=True`` rounds every GEMM operand
to bfloat16 (fp32 accumulate), mirroring the build's compute mode, so that tests
can separate precision noise from real defects; the fp32 path is the reference
semantics.
"""
import math

import numpy as np
import torch

from . import rng_ref


# --------------------------------------------------------------------------- #
# configuration / dropout-site numbering (shared convention with chambers_amd.engine)
# --------------------------------------------------------------------------- #
SITE_EMBED = 0          # Dropout after pos_embedding (vision_transformer.py:261)


def site_attn(layer):   # dropout on attention probabilities (layers/attention.py:44-46)
    return 1 + 3 * layer


def site_proj(layer):   # dropout1 (layers/transformer.py:38,69)
    return 2 + 3 * layer


def site_mlp(layer):    # dropout2 (layers/transformer.py:48,76)
    return 3 + 3 * layer


def _bf(x, on):
    return x.to(torch.bfloat16).to(torch.float32) if on else x


# ---- bf16 emulation of the BACKWARD rounding points (bf16=True only) ------------------------- #
# The build stores every GEMM / attention operand of the backward pass in bf16 as well: dz (dropout-backward of the residual
# gradient), d(pre-activation) = (dz . W2^T) * gelu'(a) with gelu' itself SAVED in bf16, dO, dqkv, d(LayerNorm output).  The
# plain `_bf` cast already rounds the gradient of a tensor it rounded in the forward (autograd of .to(bfloat16)); the functions
# below add the points a cast cannot express, so that "engine vs emulating oracle" isolates real defects from rounding.
class _RoundGrad(torch.autograd.Function):
    """identity forward; the gradient is rounded to bf16 (a gradient tensor the build stores in bf16)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _BfForwardOnly(torch.autograd.Function):
    """bf16 rounding in the forward, identity in the backward (the consumer's backward does its own rounding)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _GeluSavedDerivative(torch.autograd.Function):
    """gelu with the build's backward: d(pre-activation) = bf16(incoming fp32 accumulator * bf16(gelu'(a)))."""

    @staticmethod
    def forward(ctx, a):
        cdf = 0.5 * (1.0 + torch.erf(a / 1.4142135623730951))
        d = cdf + a * torch.exp(-0.5 * a * a) * 0.3989422804014327
        ctx.save_for_backward(d.to(torch.bfloat16).to(torch.float32))
        return a * cdf

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return (g * d).to(torch.bfloat16).to(torch.float32)


class _AttentionBf16(torch.autograd.Function):
    """Scaled dot-product attention with dropout on the probabilities, rounding points of the HIP kernels in BOTH directions:
    forward - un-normalised exponentials (times the keep mask) rounded to bf16 for P.V, 1/sum and 1/(1-rate) on the fp32 result,
    output stored bf16; backward (csrc/attention.hip, lean kernel) - probabilities recomputed from the saved log-sum-exp in fp32,
    P*keep/(1-rate) and dS = P*(dP*keep/(1-rate) - delta) rounded to bf16 as MFMA operands, delta = rowsum(dO * O) with the
    stored bf16 O, 1/sqrt(hd) applied to dQ / dK at the store, dQ / dK / dV stored bf16."""

    @staticmethod
    def forward(ctx, q, k, v, keep, inv_keep, scale):
        bf = lambda t: t.to(torch.bfloat16).to(torch.float32)   # noqa: E731
        s = torch.matmul(q, k.transpose(-1, -2)) * scale
        mx = s.max(dim=-1, keepdim=True).values
        pt = torch.exp(s - mx)
        denom = pt.sum(dim=-1, keepdim=True)
        lse = mx + torch.log(denom)
        ptd = pt * keep if keep is not None else pt
        o = bf(torch.matmul(bf(ptd), v) * (inv_keep / denom))
        ctx.save_for_backward(q, k, v, o, lse, keep if keep is not None else torch.ones(()))
        ctx.has_keep, ctx.inv_keep, ctx.scale = keep is not None, inv_keep, scale
        return o

    @staticmethod
    def backward(ctx, do):
        bf = lambda t: t.to(torch.bfloat16).to(torch.float32)   # noqa: E731
        q, k, v, o, lse, keep = ctx.saved_tensors
        do = bf(do)
        p = torch.exp(torch.matmul(q, k.transpose(-1, -2)) * ctx.scale - lse)
        keepc = (keep * ctx.inv_keep) if ctx.has_keep else 1.0
        delta = (do * o).sum(dim=-1, keepdim=True)
        dp = torch.matmul(do, v.transpose(-1, -2))
        pd = bf(p * keepc)
        ds = bf(p * (dp * keepc - delta))
        dv = bf(torch.matmul(pd.transpose(-1, -2), do))
        dk = bf(torch.matmul(ds.transpose(-1, -2), q) * ctx.scale)
        dq = bf(torch.matmul(ds, k) * ctx.scale)
        return dq, dk, dv, None, None, None


NATIVE_DROPOUT = "native"      # pass as every site's key: dropout masks from torch.rand (bench.py's cpu_baseline leg)


def _drop(x, rate, key):
    """tf.nn.dropout: x * 1/(1-rate) * keep (keras Dropout, training)."""
    if rate == 0.0 or key is None:
        return x
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    if isinstance(key, str):      # NATIVE_DROPOUT: masks from torch's generator (timing runs; no parity with the HIP path's masks)
        return x * float(scale) * (torch.rand_like(x) >= rate).to(x.dtype)
    keep = rng_ref.keep_mask(x.numel(), key, rate).reshape(tuple(x.shape))
    return x * float(scale) * torch.from_numpy(keep).to(x.dtype)


def gelu(x):
    """activations.py:46-56 (exact erf branch)."""
    return 0.5 * x * (1.0 + torch.erf(x / 1.4142135623730951))


def layer_norm(x, gamma, beta, eps):
    """keras LayerNormalization (upstream restated): biased variance over the
    last axis, gamma * (x - mean) * rsqrt(var + eps) + beta."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def multi_head_attention(x, p, prefix, num_heads, rate, key, bf16, taps=None):
    """layers/attention.py:99-127 with q = v = k = x (layers/transformer.py:66-68).  taps (dict, optional): receives the
    projected "q", "k", "v" [B,H,N,hd] and the attention output "o" (gradients retained) under prefix + name."""
    wq, bq = p[prefix + "w_query"], p[prefix + "b_query"]
    wv, bv = p[prefix + "w_value"], p[prefix + "b_value"]
    wk, bk = p[prefix + "w_key"], p[prefix + "b_key"]
    wp, bp = p[prefix + "w_projection"], p[prefix + "b_projection"]
    head_dim = wq.shape[-1]
    xb = _bf(x, bf16)
    query = torch.einsum("btd,dnh->bnth", xb, _bf(wq, bf16)) + bq
    value = torch.einsum("btd,dnh->bnth", xb, _bf(wv, bf16)) + bv
    keyt = torch.einsum("btd,dnh->bnth", xb, _bf(wk, bf16)) + bk
    if bf16:
        # operands stored bf16 by the QKV GEMM; their gradients (dqkv) are rounded by the attention backward itself
        query, value, keyt = _BfForwardOnly.apply(query), _BfForwardOnly.apply(value), _BfForwardOnly.apply(keyt)
    if taps is not None:
        for nm, t in (("q", query), ("k", keyt), ("v", value)):
            if t.requires_grad:
                t.retain_grad()
            taps[prefix + nm] = t
    if bf16:
        keep, inv_keep = None, 1.0
        if rate != 0.0 and key is not None:
            keep = torch.from_numpy(rng_ref.attn_keep_mask((x.shape[0], num_heads, x.shape[1], x.shape[1]), key, rate)).to(torch.float32)
            inv_keep = float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate)))
        attn = _AttentionBf16.apply(query, keyt, value, keep, inv_keep, 1.0 / math.sqrt(head_dim))
        if taps is not None:
            if attn.requires_grad:
                attn.retain_grad()
            taps[prefix + "o"] = attn
        return torch.einsum("bnth,ndh->btd", attn, _bf(wp, bf16)) + bp
    # ScaledAttention._calculate_scores (layers/attention.py:13-23): matmul, THEN divide
    scores = torch.matmul(query, keyt.transpose(-1, -2)) / math.sqrt(head_dim)
    weights = torch.softmax(scores, dim=-1)
    if rate != 0.0 and key is not None:     # keras Attention dropout on the probabilities; mask index: rng_ref.attn_keep_mask
        keep = (torch.rand_like(weights) >= rate) if isinstance(key, str) else torch.from_numpy(rng_ref.attn_keep_mask(tuple(weights.shape), key, rate))
        weights = weights * float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate))) * keep.to(weights.dtype)
    attn = torch.matmul(weights, value)
    if taps is not None:
        if attn.requires_grad:
            attn.retain_grad()
        taps[prefix + "o"] = attn
    return torch.einsum("bnth,ndh->btd", attn, _bf(wp, bf16)) + bp


def encoder_layer(x, p, prefix, cfg, keys, layer, bf16, taps=None):
    """EncoderLayer.call pre-norm branch, layers/transformer.py:56-58,65-77."""
    rate = cfg["dropout_rate"]
    eps = cfg.get("norm_epsilon", 1e-6)
    h = layer_norm(x, p[prefix + "norm1/gamma"], p[prefix + "norm1/beta"], eps)
    if taps is not None:
        taps[prefix + "h1"] = h
    a = multi_head_attention(h, p, prefix + "multi_head_attention/", cfg["n_heads"], rate,
                             keys.get(site_attn(layer)), bf16, taps)
    if bf16:
        a = _RoundGrad.apply(a)          # dz = bf16(dropout-backward of the residual gradient): operand of the projection's backward GEMMs
    x = x + _drop(a, rate, keys.get(site_proj(layer)))
    h = layer_norm(x, p[prefix + "norm2/gamma"], p[prefix + "norm2/beta"], eps)
    if taps is not None:
        taps[prefix + "xmid"], taps[prefix + "h2"] = x, h
    a1 = torch.matmul(_bf(h, bf16), _bf(p[prefix + "dense1/kernel"], bf16)) + p[prefix + "dense1/bias"]
    if bf16:
        u = _BfForwardOnly.apply(_GeluSavedDerivative.apply(a1))    # u stored bf16; d(a1) = bf16(fp32 accumulator * saved bf16 gelu')
    else:
        u = gelu(a1)
    y = torch.matmul(u, _bf(p[prefix + "dense2/kernel"], bf16)) + p[prefix + "dense2/bias"]
    if bf16:
        y = _RoundGrad.apply(y)          # dz of the MLP branch
    out = x + _drop(y, rate, keys.get(site_mlp(layer)))
    if taps is not None:
        taps[prefix + "u"], taps[prefix + "xout"] = u, out
    return out


def encoder_layer_post_norm(x, p, prefix, cfg, keys, layer, bf16):
    """EncoderLayer.call post-norm branch (the reference's default, pre_norm=False), layers/transformer.py:59-61,65-77:
    x = norm1(x + dropout1(attn(x))); x = norm2(x + dropout2(dense2(gelu(dense1(x)))))."""
    rate = cfg["dropout_rate"]
    eps = cfg.get("norm_epsilon", 1e-6)
    a = multi_head_attention(x, p, prefix + "multi_head_attention/", cfg["n_heads"], rate, keys.get(site_attn(layer)), bf16)
    # bf16=True mirrors the build's storage points: the LayerNorm kernel writes bf16 (in the pre-norm block that is a GEMM operand
    # anyway; here it is also the residual stream and the block's output)
    x = _bf(layer_norm(x + _drop(a, rate, keys.get(site_proj(layer))), p[prefix + "norm1/gamma"], p[prefix + "norm1/beta"], eps), bf16)
    a1 = torch.matmul(x, _bf(p[prefix + "dense1/kernel"], bf16)) + p[prefix + "dense1/bias"]
    u = _bf(gelu(a1), bf16)
    y = torch.matmul(u, _bf(p[prefix + "dense2/kernel"], bf16)) + p[prefix + "dense2/bias"]
    return _bf(layer_norm(x + _drop(y, rate, keys.get(site_mlp(layer))), p[prefix + "norm2/gamma"], p[prefix + "norm2/beta"], eps), bf16)


def patch_embed(images, kernel, bias, patch, bf16):
    """Conv2D(D, kernel=p, stride=p, 'valid') + Reshape([-1, D])
    (vision_transformer.py:235-248); kernel is HWIO [p, p, C, D]."""
    b, h, w, c = images.shape
    gh, gw = h // patch, w // patch
    x = images[:, :gh * patch, :gw * patch, :].reshape(b, gh, patch, gw, patch, c)
    x = x.permute(0, 1, 3, 2, 4, 5).reshape(b, gh * gw, patch * patch * c)
    k2 = kernel.reshape(patch * patch * c, -1)
    return torch.matmul(_bf(x, bf16), _bf(k2, bf16)) + bias


def vit_forward(p, images, cfg, keys=None, bf16=False, return_tokens=False, taps=None):
    """VisionTransformer graph, vision_transformer.py:235-283.
    ``images``: float32 NHWC, already normalised.  ``keys``: {site: key} for
    training-mode dropout, None/{} for inference.  Returns logits (or the
    pooled/feature vector when the model has no top)."""
    keys = keys or {}
    rate = cfg["dropout_rate"]
    x = patch_embed(images, p["patch_embeddings/embedding/kernel"],
                    p["patch_embeddings/embedding/bias"], cfg["patch_size"], bf16)
    b = x.shape[0]
    distilled = "add_dist_token/embeddings" in p         # DistilledVisionTransformer, vision_transformer.py:340-357
    if distilled:
        x = torch.cat([p["add_dist_token/embeddings"].unsqueeze(0).expand(b, -1, -1), x], dim=1)
    cls = p["add_cls_token/embeddings"].unsqueeze(0).expand(b, -1, -1)
    x = torch.cat([cls, x], dim=1)                       # ConcatEmbedding side="left": [cls, (dist,) patches]
    x = x + p["pos_embedding/embeddings"]                # LearnedEmbedding1D
    x = _drop(x, rate, keys.get(SITE_EMBED))
    if taps is not None:
        taps["x0"] = x
    for i in range(cfg["n_encoder_layers"]):
        x = encoder_layer(x, p, "encoder/layer_%d/" % i, cfg, keys, i, bf16, taps)
    x = layer_norm(x, p["encoder/norm/gamma"], p["encoder/norm/beta"],
                   cfg.get("norm_epsilon", 1e-6))
    if return_tokens:
        return x
    x_dist = x[:, 1, :] if distilled else None           # "dist_embedding": Cropping1D((1, N-2)), :375-381
    pooling = cfg.get("pooling", "cls")
    if pooling in ("avg", "max", "sum"):
        x = _bf(x, bf16)     # the build pools the bf16 LayerNorm output (fp32 reduction)
    if pooling == "cls":
        x = x[:, 0, :]
    elif pooling == "avg":
        x = x[:, 1:, :].mean(dim=1)
    elif pooling == "max":
        x = x[:, 1:, :].max(dim=1).values
    elif pooling == "sum":
        x = x[:, 1:, :].sum(dim=1)
    if "feature/kernel" in p:
        x = torch.tanh(torch.matmul(_bf(x, bf16), _bf(p["feature/kernel"], bf16)) + p["feature/bias"])
    if "predictions/kernel" in p:
        x = torch.matmul(_bf(x, bf16), _bf(p["predictions/kernel"], bf16)) + p["predictions/bias"]
    if distilled:
        if "predictions_dist/kernel" in p:
            x_dist = torch.matmul(_bf(x_dist, bf16), _bf(p["predictions_dist/kernel"], bf16)) + p["predictions_dist/bias"]
        if cfg.get("return_dist_token", True):           # :392-395
            return x, x_dist
        return (x + x_dist) / 2.0                        # keras Average
    return x


def sparse_ce_from_logits(logits, labels):
    """keras SparseCategoricalCrossentropy(from_logits=True), mean over the batch."""
    return torch.nn.functional.cross_entropy(logits, labels, reduction="mean")


def adamw_step(params, grads, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7,
               weight_decay=0.0, decay_mask=None):
    """optimizers.py:147-155 then keras Adam (upstream restated):
    var -= wd*var (wd NOT scaled by lr); m,v update; var -= lr_t*m/(sqrt(v)+eps),
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t).  In place on fp32 numpy-backed tensors."""
    # keras Adam keeps its hyper-parameters as float32 tensors: 1-beta, beta^t and lr_t are fp32 expressions
    b1, b2 = np.float32(beta1), np.float32(beta2)
    lr_t = np.float32(lr) * np.sqrt(np.float32(1.0) - np.power(b2, np.float32(step))) / (np.float32(1.0) - np.power(b1, np.float32(step)))
    for name in params:
        w, g = params[name], grads[name]
        if decay_mask is None or decay_mask.get(name, True):
            w.sub_(np.float32(weight_decay) * w)
        m[name].add_((g - m[name]) * float(np.float32(1.0) - b1))
        v[name].add_((g * g - v[name]) * float(np.float32(1.0) - b2))
        w.sub_(float(lr_t) * m[name] / (torch.sqrt(v[name]) + np.float32(eps)))


# --------------------------------------------------------------------------- #
# timm block, for the one equivalence the reference states
# --------------------------------------------------------------------------- #
def timm_block(x, w, num_heads, eps=1e-6):
    """timm VisionTransformer Block forward (what manual_test_vit_weights.py:252-279
    compares chambers' EncoderLayer against).  ``w`` holds timm-named tensors."""
    b, n, d = x.shape
    hd = d // num_heads
    h = torch.nn.functional.layer_norm(x, (d,), w["norm1.weight"], w["norm1.bias"], eps)
    qkv = torch.nn.functional.linear(h, w["attn.qkv.weight"], w["attn.qkv.bias"])
    qkv = qkv.reshape(b, n, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(b, n, d)
    x = x + torch.nn.functional.linear(o, w["attn.proj.weight"], w["attn.proj.bias"])
    h = torch.nn.functional.layer_norm(x, (d,), w["norm2.weight"], w["norm2.bias"], eps)
    u = torch.nn.functional.gelu(torch.nn.functional.linear(h, w["mlp.fc1.weight"], w["mlp.fc1.bias"]))
    return x + torch.nn.functional.linear(u, w["mlp.fc2.weight"], w["mlp.fc2.bias"])


def timm_to_chambers_block(w, num_heads):
    """map_encoder_layer, test_units/manual_test_vit_weights.py:27-76."""
    dim = w["attn.proj.weight"].shape[0]
    hd = dim // num_heads
    wq, wk, wv = w["attn.qkv.weight"].reshape(3, num_heads, hd, dim).permute(0, 3, 1, 2)
    bq, bk, bv = w["attn.qkv.bias"].reshape(3, num_heads, 1, hd)
    wp = w["attn.proj.weight"].reshape(dim, num_heads, hd).permute(1, 0, 2)
    return {
        "multi_head_attention/w_query": wq.contiguous(), "multi_head_attention/b_query": bq.contiguous(),
        "multi_head_attention/w_value": wv.contiguous(), "multi_head_attention/b_value": bv.contiguous(),
        "multi_head_attention/w_key": wk.contiguous(), "multi_head_attention/b_key": bk.contiguous(),
        "multi_head_attention/w_projection": wp.contiguous(),
        "multi_head_attention/b_projection": w["attn.proj.bias"].unsqueeze(0),
        "norm1/gamma": w["norm1.weight"], "norm1/beta": w["norm1.bias"],
        "dense1/kernel": w["mlp.fc1.weight"].t().contiguous(), "dense1/bias": w["mlp.fc1.bias"],
        "dense2/kernel": w["mlp.fc2.weight"].t().contiguous(), "dense2/bias": w["mlp.fc2.bias"],
        "norm2/gamma": w["norm2.weight"], "norm2/beta": w["norm2.bias"],
    }
