"""The floating-point bar of the bf16 ViT path, measured (test infrastructure: imports the CPU oracle).

    python tests/fp_bar.py [--full] [--json PATH]

For each BASELINE geometry the HIP engine runs one forward + loss + backward on seeded inputs, and every tensor family is
compared (relative L2) with
  * the plain fp32 oracle  (reference semantics; the north star's "within 1e-3 rel for bf16 attention / FFN" is judged here),
  * the oracle with the build's bf16 rounding points emulated in both directions (oracle/vit_ref.py, bf16=True),
next to the bf16 STORAGE floor: the relative L2 distance between the fp32-oracle tensor and that same tensor rounded once to
bf16 (only meaningful for tensors the build stores in bf16; ~1.6e-3 for random data: 8 significant bits).  A tensor the
build stores in bf16 cannot be closer to the fp32 oracle than that floor, whatever the kernel does.

tests/test_fp_bar_gpu.py asserts the fp32 column of the reduced-depth rows against the rounding-point BUDGET of tests/fp_budget.py
(floor x sqrt(k + kappa^2) x 1.5) and the emulated column against tight measured bounds; DESIGN.md section 2 carries the full-depth
table this script prints with --full.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import augment_ref as A      # noqa: E402
from oracle import rng_ref, vit_ref      # noqa: E402

# geometry name -> (ViTConfig kwargs, batch, train?)   depth is overridden by `depth` for the reduced (test) variants
GEOMETRIES = {
    "config1 ViT-Ti/16 224 fwd": (dict(patch_size=16, patch_dim=192, n_encoder_layers=12, n_heads=3, ff_dim=768, image_size=(224, 224)), 8, False),
    "config3 ViT-B/16 224 train": (dict(patch_size=16, patch_dim=768, n_encoder_layers=12, n_heads=12, ff_dim=3072, image_size=(224, 224)), 4, True),
    "config4 ViT-L/16 224 train": (dict(patch_size=16, patch_dim=1024, n_encoder_layers=24, n_heads=16, ff_dim=4096, image_size=(224, 224)), 2, True),
    "config5 ViT-B/16 384 train": (dict(patch_size=16, patch_dim=768, n_encoder_layers=12, n_heads=12, ff_dim=3072, image_size=(384, 384)), 2, True),
}

FAMILIES = [   # gradient families: name -> suffixes of the Keras-named weights that belong to it
    ("d patch-embedding kernel", ("patch_embeddings/embedding/kernel",)),
    ("d pos / cls embeddings", ("pos_embedding/embeddings", "add_cls_token/embeddings")),
    ("d w_query / w_key / w_value", ("w_query", "w_key", "w_value")),
    ("d b_query / b_value", ("b_query", "b_value")),
    ("d w_projection", ("w_projection",)),
    ("d dense1 kernel", ("dense1/kernel",)),
    ("d dense2 kernel", ("dense2/kernel",)),
    ("d biases (proj, dense1, dense2, patch)", ("b_projection", "dense1/bias", "dense2/bias", "patch_embeddings/embedding/bias")),
    ("d LayerNorm gamma / beta", ("gamma", "beta")),
    ("d predictions kernel / bias", ("predictions/kernel", "predictions/bias")),
]


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-300))


def bf16_floor(t):
    t = torch.as_tensor(t).float()
    return rel_l2(t.to(torch.bfloat16).float(), t)


def _weights(cfg, seed=1234):
    from chambers_amd.engine import init_keras_weights
    kw = init_keras_weights(cfg, seed=seed)
    g = np.random.Generator(np.random.PCG64(0))
    for k in kw:      # non-trivial biases / LayerNorm parameters so every term is exercised
        if k.endswith(("bias", "beta", "b_query", "b_key", "b_value", "b_projection")):
            kw[k] = g.normal(0, 0.05, size=kw[k].shape).astype(np.float32)
        if k.endswith("gamma"):
            kw[k] = (1.0 + g.normal(0, 0.1, size=kw[k].shape)).astype(np.float32)
    return kw


def measure(name, depth=None, dropout=0.1, seed=3, threads=None):
    """Returns {row: {"fp32": rel-L2 vs fp32 oracle, "emu": vs bf16-emulating oracle, "floor": bf16 storage floor or None}}."""
    from chambers_amd.engine import ViTConfig, ViTEngine
    if threads:
        torch.set_num_threads(threads)
    kwargs, bsz, train = GEOMETRIES[name]
    kwargs = dict(kwargs, dropout_rate=dropout if train else 0.0, classes=1000)
    if depth is not None:
        kwargs["n_encoder_layers"] = depth
    cfg = ViTConfig(**kwargs)
    kw = _weights(cfg)
    g = np.random.Generator(np.random.PCG64(0))
    images = g.integers(0, 256, size=(bsz,) + cfg.image_size + (3,), dtype=np.uint8)
    labels = torch.as_tensor(g.integers(0, cfg.classes, size=(bsz,)))
    eng = ViTEngine(cfg, bsz, training=train, seed=seed)
    eng.load_keras_weights(kw)
    n, h, hd, d, L = cfg.n_tokens, cfg.n_heads, cfg.head_dim, cfg.patch_dim, cfg.n_encoder_layers
    logits = eng.forward(torch.as_tensor(images, device="cuda"), training=train).float().cpu()
    got = {"logits": logits}
    last = L - 1 if train else 0      # the inference engine keeps one activation set (the last block's)

    def heads(t2d, m):                # [M, H*hd] (bf16, padded rows) -> [B, H, N, hd] fp32
        return t2d[:m].float().cpu().reshape(bsz, n, h, hd).permute(0, 2, 1, 3)

    a_last = eng.acts[last if train else 0]
    got["o (last block)"] = heads(a_last["o"], bsz * n)
    if train:
        got["o (block 0)"] = heads(eng.acts[0]["o"], bsz * n)
        got["loss per sample"] = eng.loss(labels.cuda()).float().cpu()
        eng.backward()
        dqkv = eng.dqkv[:bsz * n].float().cpu()
        got["dq (block 0)"] = dqkv[:, :d].reshape(bsz, n, h, hd).permute(0, 2, 1, 3)
        got["dk (block 0)"] = dqkv[:, d:2 * d].reshape(bsz, n, h, hd).permute(0, 2, 1, 3)
        got["dv (block 0)"] = dqkv[:, 2 * d:].reshape(bsz, n, h, hd).permute(0, 2, 1, 3)
        got["dO (block 0)"] = heads(eng.do, bsz * n)
        grads = eng.export_keras_grads()
    x = torch.from_numpy(A.imagenet_normalize(images, "tf"))
    keys = {s: rng_ref.site_key(seed, 0, s) for s in range(1 + 3 * L)} if train else None
    rows = {}
    refs = {}
    kap = {}
    rng_ref_site_attn = vit_ref.site_attn
    for mode in ("fp32", "emu"):
        p = {k: torch.tensor(v, dtype=torch.float32, requires_grad=train) for k, v in kw.items()}
        taps = {}
        ctx = torch.enable_grad() if train else torch.no_grad()
        with ctx:
            ref_logits = vit_ref.vit_forward(p, x, cfg.as_oracle_cfg(), keys=keys, bf16=(mode == "emu"), taps=taps)
            ref = {"logits": ref_logits.detach()}
            pre_last = "encoder/layer_%d/multi_head_attention/" % (L - 1)
            ref["o (last block)"] = taps[pre_last + "o"].detach()
            if train:
                ref["o (block 0)"] = taps["encoder/layer_0/multi_head_attention/o"].detach()
                per = torch.nn.functional.cross_entropy(ref_logits, labels, reduction="none")
                ref["loss per sample"] = per.detach()
                per.mean().backward()
                pre0 = "encoder/layer_0/multi_head_attention/"
                ref["dq (block 0)"], ref["dk (block 0)"], ref["dv (block 0)"] = (taps[pre0 + c].grad for c in "qkv")
                ref["dO (block 0)"] = taps[pre0 + "o"].grad
                if mode == "fp32":      # cancellation ratios of the attention backward's rounded-operand products (tests/fp_budget.py)
                    from fp_budget import attention_kappas
                    keep, inv_keep = None, 1.0
                    if kwargs["dropout_rate"]:
                        keep = torch.from_numpy(rng_ref.attn_keep_mask((bsz, h, n, n), keys[rng_ref_site_attn(0)], kwargs["dropout_rate"]))
                        inv_keep = float(np.float32(1.0) / (np.float32(1.0) - np.float32(kwargs["dropout_rate"])))
                    kap = attention_kappas(*(taps[pre0 + c].detach() for c in "qkv"), taps[pre0 + "o"].detach(), taps[pre0 + "o"].grad,
                                           keep, inv_keep)
                for fam, suffixes in FAMILIES:
                    ks = [k for k in kw if k.endswith(suffixes) and not k.endswith("b_key")]
                    if ks:
                        ref[fam] = torch.cat([p[k].grad.reshape(-1) for k in ks])
                        got[fam] = torch.cat([torch.as_tensor(grads[k]).reshape(-1) for k in ks])
        refs[mode] = ref
    stored_bf16 = {"o (last block)", "o (block 0)", "dq (block 0)", "dk (block 0)", "dv (block 0)", "dO (block 0)"}
    for row in refs["fp32"]:
        rows[row] = {"fp32": rel_l2(got[row], refs["fp32"][row]), "emu": rel_l2(got[row], refs["emu"][row]),
                     "floor": bf16_floor(refs["fp32"][row]) if row in stored_bf16 else None,
                     "kappa": kap.get(row.split(" ")[0]) if train else None}
    rows["_depth"] = L
    del eng
    torch.cuda.empty_cache()
    return rows


# reduced-depth rows asserted by tests/test_fp_bar_gpu.py: (geometry, depth) -> {row: (bound vs fp32 oracle, bound vs emulating oracle)}
# = measured on MI355X x 1.5 (python tests/fp_bar.py; the measured values are in DESIGN.md section 2)
BOUNDS = {}


def main():
    full = "--full" in sys.argv
    out = {}
    for name, (kw, _b, _t) in GEOMETRIES.items():
        depth = None if full else min(kw["n_encoder_layers"], 2)
        rows = measure(name, depth=depth)
        key = "%s, depth %d" % (name, kw["n_encoder_layers"] if depth is None else depth)
        out[key] = rows
        print("== " + key, flush=True)
        from fp_budget import budget_for_row
        L = rows.pop("_depth")
        print("   %-42s %12s %12s %12s %8s %12s" % ("tensor", "vs fp32", "vs bf16-emu", "bf16 floor", "kappa", "budget(fp32)"))
        for row, v in rows.items():
            print("   %-42s %12.3e %12.3e %12s %8s %12.3e" % (row, v["fp32"], v["emu"], "%.3e" % v["floor"] if v["floor"] is not None else "-",
                                                            "%.1f" % v["kappa"] if v.get("kappa") else "-", budget_for_row(row, L, v.get("kappa"))), flush=True)
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
