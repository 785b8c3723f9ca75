
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import fp_bar
from oracle import augment_ref as A, vit_ref, rng_ref
from chambers_amd.engine import ViTConfig, ViTEngine
kwargs, bsz, train = fp_bar.GEOMETRIES["config3 ViT-B/16 224 train"]
cfg = ViTConfig(**dict(kwargs, n_encoder_layers=2, dropout_rate=0.0, classes=1000))
kw = fp_bar._weights(cfg)
g = np.random.Generator(np.random.PCG64(0))
images = g.integers(0, 256, size=(bsz,) + cfg.image_size + (3,), dtype=np.uint8)
eng = ViTEngine(cfg, bsz, training=True, seed=3)
eng.load_keras_weights(kw)
logits = eng.forward(torch.as_tensor(images, device="cuda"), training=True).float().cpu()
x = torch.from_numpy(A.imagenet_normalize(images, "tf"))
n, d, M = cfg.n_tokens, cfg.patch_dim, bsz * cfg.n_tokens
for mode in (True, False):
    p = {k: torch.tensor(v) for k, v in kw.items()}
    taps = {}
    with torch.no_grad():
        ref = vit_ref.vit_forward(p, x, cfg.as_oracle_cfg(), keys=None, bf16=mode, taps=taps)
    r = fp_bar.rel_l2
    print("bf16-emu" if mode else "fp32")
    print("  x0", r(eng.xs[0][:M].cpu().reshape(bsz, n, d), taps["x0"]))
    for l in range(2):
        pre = "encoder/layer_%d/" % l
        a = eng.acts[l]
        print("  L%d h1 %.3e" % (l, r(a["h1"][:M].float().cpu().reshape(bsz, n, d), taps[pre + "h1"])))
        q = a["qkv"][:M].float().cpu()
        hh = cfg.n_heads
        for j, nm in enumerate("qkv"):
            print("  L%d %s %.3e" % (l, nm, r(q[:, j * d:(j + 1) * d].reshape(bsz, n, hh, 64).permute(0, 2, 1, 3), taps[pre + "multi_head_attention/" + nm])))
        print("  L%d o %.3e" % (l, r(a["o"][:M].float().cpu().reshape(bsz, n, hh, 64).permute(0, 2, 1, 3), taps[pre + "multi_head_attention/o"])))
        print("  L%d xmid %.3e" % (l, r(a["xmid"][:M].cpu().reshape(bsz, n, d), taps[pre + "xmid"])))
        print("  L%d h2 %.3e" % (l, r(a["h2"][:M].float().cpu().reshape(bsz, n, d), taps[pre + "h2"])))
        print("  L%d u %.3e" % (l, r(a["u"][:M].float().cpu().reshape(bsz, n, -1), taps[pre + "u"])))
        print("  L%d xout %.3e" % (l, r(eng.xs[l + 1][:M].cpu().reshape(bsz, n, d), taps[pre + "xout"])))
    print("  hf", r(eng.hf[:bsz].float().cpu(), torch.nn.functional.layer_norm(taps["encoder/layer_1/xout"][:, 0], (d,), p["encoder/norm/gamma"], p["encoder/norm/beta"], 1e-6)))
    print("  logits", r(logits, ref))
