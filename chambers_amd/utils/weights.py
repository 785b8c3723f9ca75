"""Weight import for the ViT builders: timm `VisionTransformer.state_dict()` -> the reference's Keras-named arrays.

The reference ships its pretrained ViT weights as Keras .h5 files converted from timm; the conversion rules are the ones its
manual equivalence test states (test_units/manual_test_vit_weights.py:27-155): conv kernel OIHW -> HWIO, fused qkv rows split
into per-head query / key / value kernels `[D, heads, hd]`, projection `[D, D]` -> `[heads, D, hd]`, `nn.Linear` weights
transposed to Keras `[in, out]`, LayerNorm weight / bias -> gamma / beta.  The result feeds `Model.assign_keras_weights` /
`ViTEngine.load_keras_weights` (numpy only; no timm or torch import is needed — any mapping of name -> array works)."""
import numpy as np


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float32)


def timm_state_dict_to_keras(state_dict, n_heads, include_top=True):
    """Returns {keras name: float32 array} for a timm ViT state dict (keys `patch_embed.proj.*`, `cls_token`, `pos_embed`,
    `blocks.<i>.*`, `norm.*`, `head.*`).  Raises KeyError on a missing tensor and ValueError on inconsistent shapes."""
    sd = {k: _np(v) for k, v in state_dict.items()}
    out = {}
    conv = sd["patch_embed.proj.weight"]                       # [D, 3, p, p]
    d = conv.shape[0]
    if d % n_heads:
        raise ValueError("embedding width %d is not divisible by %d heads" % (d, n_heads))
    hd = d // n_heads
    out["patch_embeddings/embedding/kernel"] = np.ascontiguousarray(conv.transpose(2, 3, 1, 0))   # HWIO
    out["patch_embeddings/embedding/bias"] = sd["patch_embed.proj.bias"]
    out["add_cls_token/embeddings"] = sd["cls_token"].reshape(1, d)
    out["pos_embedding/embeddings"] = sd["pos_embed"].reshape(-1, d)
    n_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(n_layers):
        t, p = "blocks.%d." % i, "encoder/layer_%d/" % i
        a = p + "multi_head_attention/"
        qkv_w, qkv_b = sd[t + "attn.qkv.weight"], sd[t + "attn.qkv.bias"]
        if qkv_w.shape != (3 * d, d):
            raise ValueError("%sattn.qkv.weight has shape %s, expected %s" % (t, qkv_w.shape, (3 * d, d)))
        w3 = qkv_w.reshape(3, n_heads, hd, d).transpose(0, 3, 1, 2)          # (q, k, v) x [D, heads, hd]
        b3 = qkv_b.reshape(3, n_heads, 1, hd)
        for j, nm in enumerate(("query", "key", "value")):
            out[a + "w_" + nm], out[a + "b_" + nm] = np.ascontiguousarray(w3[j]), np.ascontiguousarray(b3[j])
        out[a + "w_projection"] = np.ascontiguousarray(sd[t + "attn.proj.weight"].reshape(d, n_heads, hd).transpose(1, 0, 2))
        out[a + "b_projection"] = sd[t + "attn.proj.bias"].reshape(1, d)
        out[p + "norm1/gamma"], out[p + "norm1/beta"] = sd[t + "norm1.weight"], sd[t + "norm1.bias"]
        out[p + "norm2/gamma"], out[p + "norm2/beta"] = sd[t + "norm2.weight"], sd[t + "norm2.bias"]
        out[p + "dense1/kernel"], out[p + "dense1/bias"] = np.ascontiguousarray(sd[t + "mlp.fc1.weight"].T), sd[t + "mlp.fc1.bias"]
        out[p + "dense2/kernel"], out[p + "dense2/bias"] = np.ascontiguousarray(sd[t + "mlp.fc2.weight"].T), sd[t + "mlp.fc2.bias"]
    out["encoder/norm/gamma"], out["encoder/norm/beta"] = sd["norm.weight"], sd["norm.bias"]
    if "dist_token" in sd:                                     # DeiT distilled checkpoints (manual_test_vit_weights.py:90-99,140-149)
        out["add_dist_token/embeddings"] = sd["dist_token"].reshape(1, d)
    if include_top:
        out["predictions/kernel"], out["predictions/bias"] = np.ascontiguousarray(sd["head.weight"].T), sd["head.bias"]
        if "head_dist.weight" in sd:
            out["predictions_dist/kernel"], out["predictions_dist/bias"] = np.ascontiguousarray(sd["head_dist.weight"].T), sd["head_dist.bias"]
    return out
