"""Size-independent properties at the bench's full per-GPU sizes (batch 512, 224x224, ViT-B/16 shapes), where the oracle would
take minutes: involutions / idempotence of the integer augmentation ops, a checksum identity for the normalisation, softmax
row-sum and gradient-sum identities for the fused attention, moment identities for LayerNorm, and fixed points of AdamW."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, H, W = 512, 224, 224


def _images(seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)


def test_augment_involutions_idempotence_and_batch_independence():
    from chambers_amd import augmentations as aug
    x = _images()
    assert torch.equal(aug.Invert()(aug.Invert()(x)), x)                                    # involution
    p3 = aug.Posterize(3)(x)
    assert torch.equal(aug.Posterize(3)(p3), p3) and int((p3 & 0x1F).max()) == 0            # idempotent, low bits cleared
    s = aug.Solarize(128)(x)
    assert torch.equal(aug.Solarize(128)(s), s) and int(s.max()) < 128                      # x<128 keeps, else 255-x (<128): idempotent
    centers = torch.randint(0, 224, (B, 2), dtype=torch.int32, device="cuda")
    c1 = aug.CutOut(72, 128)(x, centers=centers)
    assert torch.equal(aug.CutOut(72, 128)(c1, centers=centers), c1)                        # idempotent
    eq = aug.Equalize()(x)
    ac = aug.AutoContrast()(x)
    for op, full in ((aug.Equalize(), eq), (aug.AutoContrast(), ac), (aug.Sharpness(1.72), aug.Sharpness(1.72)(x))):
        assert torch.equal(op(x[100:103].contiguous()), full[100:103])                       # per-image ops: batch slices agree
    # translate by whole pixels and back: the interior is restored exactly, the border carries the fill value
    t = aug.TranslateX(16.0, fill_value=7)
    back = t(t(x, negate=False), negate=True)
    assert torch.equal(back[:, :, 16:-16], x[:, :, 16:-16])
    r0 = aug.Rotate(0.0, fill_value=128)(x, negate=False)
    assert torch.equal(r0, x)                                                               # identity transform hits every pixel incl. the last


def test_normalisation_checksum_matches_integer_sum():
    """sum(x / 127.5 - 1) over the batch == (sum(x) / 127.5 - count) up to fp32 accumulation: a checksum of checksums."""
    from chambers_amd import augmentations as aug
    x = _images(1)
    y = aug.ImageNetNormalization("tf")(x)
    assert y.dtype == torch.float32 and tuple(y.shape) == tuple(x.shape)
    per_image = y.double().sum(dim=(1, 2, 3)).cpu().numpy()
    ints = x.to(torch.int64).sum(dim=(1, 2, 3)).cpu().numpy()
    expect = ints / 127.5 - H * W * 3
    assert np.allclose(per_image, expect, rtol=0, atol=2e-2)                                # 150528 terms of <= 6e-8 error each
    assert float(y.min()) >= -1.0 and float(y.max()) <= 1.0


def test_attention_row_sum_and_gradient_sum_identities():
    """With V = 1 every output element is sum_k P[q,k] = 1; with dO = 1 and no dropout, sum_q dV[q,:] = N per head column
    (P's columns summed over queries, then over keys) and dQ = dK = 0 (dP is constant along keys, so dS = P*(dP - delta) = 0)."""
    from chambers_amd import kernels as K
    n, h = 197, 12
    d = h * 64
    g = torch.Generator(device="cuda").manual_seed(2)
    qkv = (torch.randn(B * n, 3 * d, device="cuda", generator=g)).to(torch.bfloat16)
    qkv[:, 2 * d:] = 1.0
    o = torch.empty(B * n, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B * h * n, device="cuda")
    K.attention_fwd(qkv, o, lse, B, n, h, 64)
    assert float((o.float() - 1.0).abs().max()) <= 2 ** -7                                  # bf16 P rounding, fp32 normalisation
    do = torch.ones(B * n, d, dtype=torch.bfloat16, device="cuda")
    dqkv = torch.empty(B * n, 3 * d, dtype=torch.bfloat16, device="cuda")
    K.attention_bwd(qkv, o, do, lse, dqkv, B, n, h, 64)
    dv = dqkv[:, 2 * d:].float().reshape(B, n, d).sum(dim=1)
    assert float((dv - n).abs().max()) < 0.02 * n                                           # bf16 storage of dV entries
    scale = float(dqkv[:, 2 * d:].float().abs().mean())
    assert float(dqkv[:, :2 * d].float().abs().max()) < 2e-2 * scale * n ** 0.5             # dQ, dK vanish (rounding noise only)


def test_layernorm_moments_and_adamw_fixed_points():
    from chambers_amd import kernels as K
    m, d = B * 197, 768
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(m, d, device="cuda", generator=g) * 3 + 1.5
    y = torch.empty(m, d, dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.empty(m, device="cuda"), torch.empty(m, device="cuda")
    K.layernorm_fwd(x, d, torch.ones(d, device="cuda"), torch.zeros(d, device="cuda"), y, mean, rstd, m, d, 1e-6)
    yf = y.float()
    assert float(yf.mean(dim=1).abs().max()) < 5e-3 and float((yf.var(dim=1, unbiased=False) - 1).abs().max()) < 2e-2
    assert torch.allclose(mean, x.mean(dim=1), atol=1e-5) and torch.allclose(rstd, torch.rsqrt(x.var(dim=1, unbiased=False) + 1e-6), rtol=1e-5)
    # AdamW on the ViT-B/16 parameter count: zero gradient and zero decay is a fixed point; zero gradient with decay is p*(1-wd)
    nflat = 86_568_960
    p = torch.randn(nflat, device="cuda", generator=g)
    p0 = p.clone()
    gr, mo, vo = (torch.zeros(nflat, device="cuda") for _ in range(3))
    flags = torch.ones(nflat // 1024, dtype=torch.uint8, device="cuda")
    K.adamw(p, gr, mo, vo, flags, 1e-3, 0.9, 0.999, 1e-7, 0.0, 1.0)
    assert torch.equal(p, p0) and float(mo.abs().max()) == 0.0
    flags[::2] = 0
    K.adamw(p, gr, mo, vo, flags, 1e-3, 0.9, 0.999, 1e-7, 0.25, 1.0)
    pv, p0v = p.view(-1, 1024), p0.view(-1, 1024)
    assert torch.equal(pv[::2], p0v[::2]) and torch.equal(pv[1::2], p0v[1::2] - np.float32(0.25) * p0v[1::2])
