"""GPU parity of the ViT-block kernels, called through the C ABI (chambers_amd.kernels), against
torch-CPU fp32/fp64 restatements on the same (bf16-rounded) inputs.

Tolerances (north star: 1e-3 relative for bf16 attention / FFN):
  * kernels with an fp32 output are held to rel-L2 <= 1e-3 against the fp64 result computed from the
    SAME bf16 operands (in practice ~1e-6: fp32 accumulation) ;
  * the bf16-output variant must equal the fp32-output variant rounded once to bf16 (bit-exact), so
    its only extra error is the final storage rounding (<= 2^-9 relative per element);
  * attention rounds the probabilities to bf16 before P.V: rel-L2 <= 1e-3 on the fp32 oracle result
    is asserted on bf16 outputs against a bf16-rounded oracle with atol = 1 bf16 ulp of max|ref|.
"""
import math

import numpy as np
import pytest
import torch

from oracle import rng_ref

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf(x):
    return x.to(torch.bfloat16)


def g(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------------ RNG
def test_dropout_mask_matches_rng_ref():
    from chambers_amd import kernels as K, rng
    for n, rate, key in ((1, 0.1, 7), (1001, 0.1, 0xDEADBEEF), (4096, 0.5, 123), (65537, 0.25, 0xFFFFFFFF)):
        got = K.dropout_mask(n, rate, key).cpu().numpy().astype(bool)
        np.testing.assert_array_equal(got, rng_ref.keep_mask(n, key, rate))
    assert rng.site_key(5, 3, 11) == rng_ref.site_key(5, 3, 11)
    big = K.dropout_mask(1 << 24, 0.1, 99).float().mean().item()
    assert abs(big - 0.9) < 1e-3


# ------------------------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 384, 192), (197 * 3, 768, 768), (100, 1000, 128), (1, 4, 64), (513, 132, 320)])
def test_gemm_nt_plain_and_bias(m, n, k):
    from chambers_amd import kernels as K
    a = bf(torch.randn(m, k, generator=g(1)))
    b = bf(torch.randn(n, k, generator=g(2)))
    bias = torch.randn(n, generator=g(3))
    ref = a.double() @ b.double().t() + bias.double()
    out32 = torch.empty(m, n, dtype=torch.float32, device="cuda")
    K.gemm_nt(a.cuda(), b.cuda(), out32, bias=bias.cuda())
    assert rel_l2(out32.cpu(), ref) < 1e-3
    assert rel_l2(out32.cpu(), ref) < 2e-6
    out16 = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    K.gemm_nt(a.cuda(), b.cuda(), out16, bias=bias.cuda())
    assert torch.equal(out16, out32.to(torch.bfloat16))
    # A = I with an ASYMMETRIC B catches a transposed C write
    if m == k:
        eye = bf(torch.eye(m))
        o = torch.empty(m, n, dtype=torch.float32, device="cuda")
        K.gemm_nt(eye.cuda(), b.cuda(), o)
        assert torch.equal(o.cpu(), b.float().t().contiguous())


def test_gemm_nt_padded_rows_untouched_and_strides():
    from chambers_amd import kernels as K
    m, n, k = 197, 192, 128
    a = bf(torch.randn(256, k, generator=g(4))).cuda()
    b = bf(torch.randn(n, k, generator=g(5))).cuda()
    out = torch.full((256, n), 7.0, dtype=torch.float32, device="cuda")
    K.gemm_nt(a, b, out, m=m)
    assert torch.all(out[m:] == 7.0)
    ref = a[:m].double().cpu() @ b.double().cpu().t()
    assert rel_l2(out[:m].cpu(), ref) < 2e-6


def test_gemm_nt_gelu_and_dgelu():
    from chambers_amd import kernels as K
    m, n, k = 300, 256, 128
    a = bf(torch.randn(m, k, generator=g(6)))
    b = bf(torch.randn(n, k, generator=g(7)) * 0.2)
    bias = torch.randn(n, generator=g(8)) * 0.1
    pre = a.double() @ b.double().t() + bias.double()
    ref = 0.5 * pre * (1 + torch.erf(pre / math.sqrt(2.0)))
    out = torch.empty(m, n, dtype=torch.float32, device="cuda")
    aux = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    K.gemm_nt(a.cuda(), b.cuda(), out, bias=bias.cuda(), epilogue=K.EPI_GELU, aux=aux)
    assert rel_l2(out.cpu(), ref) < 1e-5
    dref = 0.5 * (1 + torch.erf(pre / math.sqrt(2.0))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
    assert rel_l2(aux.float().cpu(), dref) < 3e-3          # gelu'(x) saved in bf16 for the backward epilogue
    assert float((aux.float().cpu() - dref).abs().max()) <= 2 ** -7
    # backward epilogue: C = acc * aux
    dy = bf(torch.randn(m, k, generator=g(9)))
    acc = dy.double() @ b.double().t()
    out2 = torch.empty(m, n, dtype=torch.float32, device="cuda")
    K.gemm_nt(dy.cuda(), b.cuda(), out2, epilogue=K.EPI_DGELU, aux=aux)
    assert rel_l2(out2.cpu(), acc * aux.double().cpu()) < 1e-5
    assert rel_l2(out2.cpu(), acc * dref) < 3e-3


@pytest.mark.parametrize("rate", [0.0, 0.1])
def test_gemm_nt_residual_dropout(rate):
    from chambers_amd import kernels as K
    m, n, k = 197 * 2, 192, 192
    a = bf(torch.randn(m, k, generator=g(10)))
    b = bf(torch.randn(n, k, generator=g(11)) * 0.1)
    bias = torch.randn(n, generator=g(12))
    resid = torch.randn(m, n, generator=g(13))
    key = 0xABCDEF
    y = a.double() @ b.double().t() + bias.double()
    if rate:
        keep = torch.from_numpy(rng_ref.keep_mask(m * n, key, rate).reshape(m, n))
        y = y * float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate))) * keep
    ref = resid.double() + y
    out = torch.empty(m, n, dtype=torch.float32, device="cuda")
    K.gemm_nt(a.cuda(), b.cuda(), out, bias=bias.cuda(), epilogue=K.EPI_RESID, resid=resid.cuda(), drop_rate=rate, drop_key=key)
    assert rel_l2(out.cpu(), ref) < 2e-6
    # in place on the residual buffer
    r2 = resid.cuda().clone()
    K.gemm_nt(a.cuda(), b.cuda(), r2, bias=bias.cuda(), epilogue=K.EPI_RESID, resid=r2, drop_rate=rate, drop_key=key)
    assert torch.equal(r2, out)


def test_gemm_nt_patch_epilogue_and_cls_row():
    from chambers_amd import kernels as K
    bsz, npatch, k, d = 3, 12, 192, 128
    a = bf(torch.randn(bsz * npatch, k, generator=g(14)))
    w = bf(torch.randn(d, k, generator=g(15)) * 0.1)
    bias = torch.randn(d, generator=g(16))
    pos = torch.randn(npatch + 1, d, generator=g(17))
    cls = torch.randn(d, generator=g(18))
    rate, key = 0.1, 4242
    x = torch.zeros(bsz * (npatch + 1), d, dtype=torch.float32, device="cuda")
    K.gemm_nt(a.cuda(), w.cuda(), x, bias=bias.cuda(), epilogue=K.EPI_PATCH, resid=pos.cuda(), period=npatch, drop_rate=rate, drop_key=key)
    K.cls_row(x, cls.cuda(), pos.cuda(), bsz, npatch + 1, d, rate, key)
    tok = (a.double() @ w.double().t() + bias.double()).reshape(bsz, npatch, d)
    full = torch.cat([cls.double().expand(bsz, 1, d), tok], dim=1) + pos.double()
    keep = torch.from_numpy(rng_ref.keep_mask(full.numel(), key, rate).reshape(full.shape))
    ref = full * float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate))) * keep
    assert rel_l2(x.cpu().reshape(bsz, npatch + 1, d), ref) < 2e-6


@pytest.mark.parametrize("m,n,k", [(4096, 512, 128), (300, 256, 64), (2500, 320, 192)])
def test_gemm_nt_fused_column_sums(m, n, k):
    """out_colsum: the consumer layer's bias gradient, fused into the epilogue (persistent kernel) or added by the
    stand-alone pass (small shapes) — same contract either way."""
    from chambers_amd import kernels as K
    a = bf(torch.randn(m, k, generator=g(25)))
    b = bf(torch.randn(n, k, generator=g(26)) * 0.1)
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    cs = torch.full((n,), 3.0, device="cuda")
    K.gemm_nt(a.cuda(), b.cuda(), out, colsum=cs)
    ref = (a.double() @ b.double().t())
    assert rel_l2(out.float().cpu(), ref) < 4e-3
    assert rel_l2((cs - 3.0).cpu(), ref.sum(0)) < 2e-3       # sums of fp32 (fused) or bf16-rounded (stand-alone) outputs


# the persistent 256x256 kernel (M >= 2048, N >= 256) is what the headline bench runs; every fused epilogue of it at sizes that
# take it, both the FAST instantiation (M and N multiples of 256: clamp-free staging, unguarded epilogue) and ragged shapes
PERSISTENT_SHAPES = [(4096, 768, 768), (4096, 3072, 768), (2304, 512, 128), (2500, 328, 192), (4096 + 77, 768, 256)]


@pytest.mark.parametrize("m,n,k", PERSISTENT_SHAPES)
@pytest.mark.parametrize("out_f32", [False, True])
def test_persistent_gemm_gelu_and_dgelu_colsum(m, n, k, out_f32):
    from chambers_amd import kernels as K
    a = bf(torch.randn(m, k, generator=g(61))).cuda()
    b = bf(torch.randn(n, k, generator=g(62)) * 0.05).cuda()
    bias = (torch.randn(n, generator=g(63)) * 0.1).cuda()
    pre = a.double().cpu() @ b.double().cpu().t() + bias.double().cpu()
    ref = 0.5 * pre * (1 + torch.erf(pre / math.sqrt(2.0)))
    dref = 0.5 * (1 + torch.erf(pre / math.sqrt(2.0))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
    odt = torch.float32 if out_f32 else torch.bfloat16
    out = torch.full((m + 2, n), 9.0, dtype=odt, device="cuda")
    aux = torch.full((m + 2, n), 5.0, dtype=torch.bfloat16, device="cuda")
    K.gemm_nt(a, b, out, m=m, bias=bias, epilogue=K.EPI_GELU, aux=aux)
    assert rel_l2(out[:m].float().cpu(), ref) < (1e-5 if out_f32 else 3e-3)
    assert float((aux[:m].float().cpu() - dref).abs().max()) <= 2 ** -7
    assert bool((out[m:].float() == 9.0).all()) and bool((aux[m:].float() == 5.0).all())
    # backward epilogue: (dY . W) * gelu' with the column sums (dense1's bias gradient) riding along
    dy = bf(torch.randn(m, k, generator=g(64))).cuda()
    acc = dy.double().cpu() @ b.double().cpu().t()
    want = acc * aux[:m].double().cpu()
    out2 = torch.full((m + 2, n), 9.0, dtype=odt, device="cuda")
    cs = torch.full((n,), 2.0, device="cuda")
    K.gemm_nt(dy, b, out2, m=m, epilogue=K.EPI_DGELU, aux=aux, colsum=cs)
    assert rel_l2(out2[:m].float().cpu(), want) < (1e-5 if out_f32 else 3e-3)
    assert rel_l2((cs - 2.0).cpu(), want.sum(0)) < 1e-4
    assert bool((out2[m:].float() == 9.0).all())


@pytest.mark.parametrize("m,n,k,epi", [(8192, 1024, 768, "resid"), (8192 + 100, 1280, 640, "none"), (16384, 768, 1024, "gelu"),
                                         (4096, 512, 448, "none")])
def test_persistent_gemm_schedules_agree_bit_for_bit(m, n, k, epi):
    """The same K order and accumulation order in every schedule of the persistent NT kernels: static tile shares vs the per-XCD
    tile queue (K >= 8 K-tiles; 70 launches per kernel so that every queue slot is used and must have been left clean), and the
    pipelined (default) vs lockstep vs ping-pong K-loops (ping-pong: full tiles; falls back to lockstep on the ragged shape)."""
    from chambers_amd import _lib
    from chambers_amd import kernels as K
    a = bf(torch.randn(m, k, generator=g(71))).cuda()
    b = bf(torch.randn(n, k, generator=g(72)) * 0.05).cuda()
    bias = (torch.randn(n, generator=g(73)) * 0.1).cuda()
    resid = torch.randn(m, n, generator=g(74)).cuda() if epi == "resid" else None
    kw = dict(bias=bias, epilogue={"resid": K.EPI_RESID, "none": K.EPI_NONE, "gelu": K.EPI_GELU}[epi], resid=resid,
              drop_rate=0.1 if epi == "resid" else 0.0, drop_key=9)

    def run():
        out = torch.full((m, n), float("nan"), dtype=torch.float32 if epi == "resid" else torch.bfloat16, device="cuda")
        aux = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda") if epi == "gelu" else None
        K.gemm_nt(a, b, out, aux=aux, **kw)
        torch.cuda.synchronize()
        return out, aux

    try:
        _lib.set_option("GEMM_TILE_QUEUE", 0)
        _lib.set_option("GEMM_ALGO", 0)
        ref, ref_aux = run()
        assert not torch.isnan(ref.float()).any()
        _lib.set_option("GEMM_TILE_QUEUE", 1)       # per-XCD tile queue: the pipelined kernel (automatic), then the lockstep one
        for algo, reps in ((0, 70), (2, 70)):
            _lib.set_option("GEMM_ALGO", algo)
            for _ in range(reps):
                out, aux = run()
                assert torch.equal(out, ref) and (aux is None or torch.equal(aux, ref_aux))
        _lib.set_option("GEMM_TILE_QUEUE", 0)
        for algo in (2, 4, 5):       # lockstep (static shares); ping-pong; pipelined reads + spread staging (the default, = ref)
            _lib.set_option("GEMM_ALGO", algo)
            for _ in range(3):
                out, aux = run()
                assert torch.equal(out, ref) and (aux is None or torch.equal(aux, ref_aux))
    finally:
        _lib.set_option("GEMM_ALGO", 0)
        _lib.set_option("GEMM_TILE_QUEUE", 0)


@pytest.mark.parametrize("m,n,k", PERSISTENT_SHAPES)
@pytest.mark.parametrize("rate", [0.0, 0.1])
def test_persistent_gemm_residual_dropout(m, n, k, rate):
    from chambers_amd import kernels as K
    a = bf(torch.randn(m, k, generator=g(65))).cuda()
    b = bf(torch.randn(n, k, generator=g(66)) * 0.1).cuda()
    bias = torch.randn(n, generator=g(67)).cuda()
    resid = torch.randn(m, n, generator=g(68))
    key = 0x13579B
    y = a.double().cpu() @ b.double().cpu().t() + bias.double().cpu()
    if rate:
        keep = torch.from_numpy(rng_ref.keep_mask(m * n, key, rate).reshape(m, n))
        y = y * float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate))) * keep
    ref = resid.double() + y
    out = torch.full((m + 2, n), 9.0, dtype=torch.float32, device="cuda")
    K.gemm_nt(a, b, out, m=m, bias=bias, epilogue=K.EPI_RESID, resid=resid.cuda(), drop_rate=rate, drop_key=key)
    assert rel_l2(out[:m].cpu(), ref) < 2e-6
    assert bool((out[m:] == 9.0).all())
    r2 = torch.cat([resid, torch.full((2, n), 9.0)]).cuda()          # in place on the residual buffer (what the engine does)
    K.gemm_nt(a, b, r2, m=m, bias=bias, epilogue=K.EPI_RESID, resid=r2, drop_rate=rate, drop_key=key)
    assert torch.equal(r2, out)


@pytest.mark.parametrize("bsz,npatch,k,d,nspecial", [(12, 196, 768, 768, 1), (11, 196, 128, 512, 1), (24, 100, 192, 256, 2)])
@pytest.mark.parametrize("rate", [0.0, 0.1])
def test_persistent_gemm_patch_epilogue(bsz, npatch, k, d, nspecial, rate):
    """EPI_PATCH at M = B * patches >= 2048: + bias + positional row, dropout on the [cls (, dist), patches] index, rows remapped
    past the special tokens (vision_transformer.py:235-261)."""
    from chambers_amd import kernels as K
    m = bsz * npatch
    assert m >= 2048
    n_tok = npatch + nspecial
    a = bf(torch.randn(m, k, generator=g(70))).cuda()
    w = bf(torch.randn(d, k, generator=g(71)) * 0.1).cuda()
    bias = torch.randn(d, generator=g(72)).cuda()
    pos = torch.randn(n_tok, d, generator=g(73))
    key = 777
    x = torch.full((bsz * n_tok, d), 9.0, dtype=torch.float32, device="cuda")
    K.gemm_nt(a, w, x, m=m, bias=bias, epilogue=K.EPI_PATCH, resid=pos.cuda(), period=npatch | ((nspecial - 1) << 24), drop_rate=rate, drop_key=key)
    tok = (a.double().cpu() @ w.double().cpu().t() + bias.double().cpu()).reshape(bsz, npatch, d) + pos.double()[nspecial:]
    got = x.cpu().reshape(bsz, n_tok, d)
    if rate:
        keep = torch.from_numpy(rng_ref.keep_mask(bsz * n_tok * d, key, rate).reshape(bsz, n_tok, d))[:, nspecial:]
        tok = tok * float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate))) * keep
    assert rel_l2(got[:, nspecial:], tok) < 2e-6
    assert bool((got[:, :nspecial] == 9.0).all())                 # the special-token rows belong to chb_token_row


def test_softmax_ce_guards_out_of_range_labels():
    from chambers_amd import kernels as K
    logits = torch.randn(6, 16, generator=g(80)).cuda()
    labels = torch.tensor([3, -1, 15, 16, 0, 1000], dtype=torch.int32, device="cuda")
    loss = torch.zeros(6, device="cuda")
    dl = torch.zeros(6, 16, dtype=torch.bfloat16, device="cuda")
    K.softmax_ce(logits, labels, loss, dl, 10, 1.0)
    ok = torch.tensor([True, False, False, False, True, False])
    assert bool(torch.isfinite(loss.cpu()[ok]).all()) and bool(torch.isnan(loss.cpu()[~ok]).all())
    assert bool(torch.isnan(dl.float().cpu()[~ok][:, :10]).all()) and bool(torch.isfinite(dl.float().cpu()[ok]).all())
    ref = torch.nn.functional.cross_entropy(logits.cpu()[ok][:, :10], labels.cpu()[ok].long(), reduction="none")
    assert torch.allclose(loss.cpu()[ok], ref, atol=1e-5)


def test_layernorm_bwd_zero_gaps_and_adamw_zero_grad():
    """The two fills folded into kernels: the class-row LayerNorm backward zeroes the rows between the class rows, and AdamW
    clears each gradient element behind its read."""
    from chambers_amd import kernels as K
    bsz, n, d = 6, 50, 192
    x = torch.randn(bsz * n, d, generator=g(81)).cuda()
    gamma = torch.randn(d, generator=g(82)).cuda()
    dy = bf(torch.randn(bsz, d, generator=g(83))).cuda()
    xr = x.view(bsz, n, d)[:, 0, :]
    mean = xr.mean(-1).contiguous()
    rstd = (1.0 / torch.sqrt(xr.var(-1, unbiased=False) + 1e-6)).contiguous()
    dx_ref = torch.zeros(bsz * n, d, device="cuda")
    dg1, db1, dg2, db2 = (torch.zeros(d, device="cuda") for _ in range(4))
    K.layernorm_bwd(dy, x, n * d, mean, rstd, gamma, dx_ref, n * d, False, dg1, db1, bsz, d)
    dx = torch.full((bsz * n + 3, d), 7.0, device="cuda")
    K.layernorm_bwd(dy, x, n * d, mean, rstd, gamma, dx, n * d, False, dg2, db2, bsz, d, zero_gaps=True)
    assert torch.equal(dx[:bsz * n], dx_ref)
    assert bool((dx[bsz * n:] == 7.0).all())
    with pytest.raises(ValueError):
        K.layernorm_bwd(dy, x, n * d, mean, rstd, gamma, dx, n * d, True, dg2, db2, bsz, d, zero_gaps=True)
    nel = 4096 * 3
    p0 = torch.randn(nel, generator=g(84))
    gr = torch.randn(nel, generator=g(85))
    flags = torch.ones(nel // 1024, dtype=torch.uint8, device="cuda")
    outs = []
    for zg in (False, True):
        p, gg, m, v = p0.cuda().clone(), gr.cuda().clone(), torch.zeros(nel, device="cuda"), torch.zeros(nel, device="cuda")
        K.adamw(p, gg, m, v, flags, 1e-3, 0.9, 0.999, 1e-7, 0.05, 1.0, zero_grad=zg)
        outs.append((p.clone(), m.clone(), v.clone()))
        assert bool((gg == 0).all()) if zg else torch.equal(gg.cpu(), gr)
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    z = torch.randn(1024 * 5, generator=g(86)).cuda()
    K.zero_f32(z)
    assert bool((z == 0).all())


@pytest.mark.parametrize("seed", range(12))
def test_gemm_shape_fuzz(seed):
    """Seeded random shapes through both GEMM forms: tile edges in M and N, 1-13 K-tiles, the persistent 256x256 kernels (M >=
    2048 / 4096) and the small-shape kernels, plain / GELU / residual epilogues, bf16 and fp32 outputs, partial-plane and atomic
    weight gradients — against a float64 product of the same bf16 operands."""
    from chambers_amd import kernels as K
    r = np.random.default_rng(1000 + seed)
    m = int(r.choice([r.integers(1, 300), r.integers(2048, 6000), r.integers(4096, 9000)]))
    n = int(r.integers(1, 140)) * 8
    k = int(r.integers(1, 14)) * 64
    a = bf(torch.randn(m, k, generator=g(200 + seed))).cuda()
    b = bf(torch.randn(n, k, generator=g(300 + seed)) * 0.1).cuda()
    bias = torch.randn(n, generator=g(400 + seed)).cuda()
    ref = a.double().cpu() @ b.double().cpu().t() + bias.double().cpu()
    epi = int(r.integers(0, 3))
    out_f32 = bool(r.integers(0, 2))
    out = torch.full((m + 3, n), 9.0, dtype=torch.float32 if out_f32 else torch.bfloat16, device="cuda")
    if epi == 0:
        K.gemm_nt(a, b, out, m=m, bias=bias)
    elif epi == 1:
        aux = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        K.gemm_nt(a, b, out, m=m, bias=bias, epilogue=K.EPI_GELU, aux=aux)
        ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2.0)))
    else:
        resid = torch.randn(m, n, generator=g(500 + seed)).cuda()
        out = torch.full((m + 3, n), 9.0, dtype=torch.float32, device="cuda")
        K.gemm_nt(a, b, out, m=m, bias=bias, epilogue=K.EPI_RESID, resid=resid)
        ref = ref + resid.double().cpu()
        out_f32 = True
    assert rel_l2(out[:m].float().cpu(), ref) < (1e-5 if out_f32 else 4e-3), (m, n, k, epi)
    assert bool((out[m:].float() == 9.0).all()), "rows past M were written"
    # weight gradient of the same operands: dW[k, n] = A^T . (A-shaped dY) needs M % 64 == 0
    mp = (m + 63) // 64 * 64
    x = torch.zeros(mp, k, dtype=torch.bfloat16, device="cuda")
    x[:m] = a
    dy = torch.zeros(mp, n, dtype=torch.bfloat16, device="cuda")
    dy[:m] = bf(torch.randn(m, n, generator=g(600 + seed))).cuda()
    wref = x.double().cpu().t() @ dy.double().cpu()
    for ws in (None, torch.empty(K.tn_workspace_elems(k, n), device="cuda")):
        dw = torch.zeros(k, n, device="cuda")
        cs = torch.zeros(n, device="cuda")
        K.gemm_tn(x, dy, dw, m=mp, ws=ws, colsum=cs)
        assert rel_l2(dw.cpu(), wref) < 5e-6, (mp, k, n, ws is None)
        assert rel_l2(cs.cpu(), dy.double().cpu().sum(0)) < 2e-5, (mp, k, n, ws is None)


def test_gemm_nt_rejects_bad_shapes():
    from chambers_amd import kernels as K
    a = torch.zeros(8, 48, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros(8, 48, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        K.gemm_nt(a, b, torch.empty(8, 8, dtype=torch.float32, device="cuda"))      # K % 64
    with pytest.raises(ValueError):
        K.gemm_nt(a.float(), b, torch.empty(8, 8, dtype=torch.float32, device="cuda"))


# ------------------------------------------------------------------------------------ GEMM TN
@pytest.mark.parametrize("m,kd,nd", [(64, 128, 128), (256, 192, 576), (1024, 768, 256), (197 * 64, 192, 768), (128, 200, 72)])
def test_gemm_tn(m, kd, nd):
    from chambers_amd import kernels as K
    x = bf(torch.randn(m, kd, generator=g(20)))
    dy = bf(torch.randn(m, nd, generator=g(21)))
    ref = x.double().t() @ dy.double()
    dw = torch.zeros(kd, nd, dtype=torch.float32, device="cuda")
    K.gemm_tn(x.cuda(), dy.cuda(), dw)
    assert rel_l2(dw.cpu(), ref) < 5e-6
    K.gemm_tn(x.cuda(), dy.cuda(), dw)           # accumulates
    assert rel_l2(dw.cpu(), 2 * ref) < 5e-6
    # asymmetric identity check: X = [I; 0] -> dW = dY[:kd]
    if m >= kd and kd % 64 == 0:
        xi = torch.zeros(m, kd)
        xi[:kd] = torch.eye(kd)
        o = torch.zeros(kd, nd, dtype=torch.float32, device="cuda")
        K.gemm_tn(bf(xi).cuda(), dy.cuda(), o)
        assert torch.equal(o.cpu(), dy[:kd].float())


@pytest.mark.parametrize("m,kd,nd", [(8192, 768, 768), (197 * 64, 192, 768), (4096, 200, 328), (12800, 768, 3072)])
def test_gemm_tn_partial_planes_match_the_atomic_epilogue(m, kd, nd):
    """chb_gemm_tn_ws: split-K partial planes in caller scratch + fold launch == the atomic epilogue (same products, another
    summation order), accumulates into dW, falls back to atomics when the scratch is too small, stays inside its planes."""
    from chambers_amd import kernels as K
    x = bf(torch.randn(m, kd, generator=g(22))).cuda()
    dy = bf(torch.randn(m, nd, generator=g(23))).cuda()
    ref = x.double().cpu().t() @ dy.double().cpu()
    need = K.tn_workspace_elems(kd, nd)
    ws = torch.full((need + 1024,), float("nan"), device="cuda")
    dw = torch.ones(kd, nd, dtype=torch.float32, device="cuda")
    K.gemm_tn(x, dy, dw, ws=ws)
    assert rel_l2(dw.cpu() - 1.0, ref) < 5e-6
    assert bool(torch.isnan(ws[need:]).all())                     # nothing written past the planes
    K.gemm_tn(x, dy, dw, ws=ws)                                   # accumulates
    assert rel_l2(dw.cpu() - 1.0, 2 * ref) < 5e-6
    atom = torch.zeros(kd, nd, dtype=torch.float32, device="cuda")
    K.gemm_tn(x, dy, atom)
    assert rel_l2(atom.cpu(), ref) < 5e-6
    # column sums of dY ride along (ones^T . dY): same products as the weight gradient, with either epilogue
    for w in (ws, None):
        cs = torch.full((nd,), 2.0, device="cuda")
        o = torch.zeros(kd, nd, dtype=torch.float32, device="cuda")
        K.gemm_tn(x, dy, o, ws=w, colsum=cs)
        assert rel_l2(o.cpu(), ref) < 5e-6
        assert rel_l2((cs - 2.0).cpu(), dy.double().cpu().sum(0)) < 5e-6
    small = torch.full((16,), float("nan"), device="cuda")        # too small: atomic epilogue, scratch untouched
    o = torch.zeros(kd, nd, dtype=torch.float32, device="cuda")
    K.gemm_tn(x, dy, o, ws=small)
    assert rel_l2(o.cpu(), ref) < 5e-6 and bool(torch.isnan(small).all())


# ------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("m,d", [(7, 192), (197 * 2, 768), (33, 1024), (5, 384), (1, 64)])
def test_layernorm_fwd_bwd(m, d):
    from chambers_amd import kernels as K
    x = torch.randn(m, d, generator=g(30)) * 2 + 0.5
    gamma = torch.randn(d, generator=g(31))
    beta = torch.randn(d, generator=g(32))
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (d,), gd, bd, 1e-6)
    y = torch.empty(m, d, dtype=torch.bfloat16, device="cuda")
    mean = torch.empty(m, dtype=torch.float32, device="cuda")
    rstd = torch.empty(m, dtype=torch.float32, device="cuda")
    K.layernorm_fwd(x.cuda(), d, gamma.cuda(), beta.cuda(), y, mean, rstd, m, d, 1e-6)
    assert torch.equal(y.cpu(), bf(ref.detach().float())) or rel_l2(y.float().cpu(), ref.detach()) < 3e-3
    assert rel_l2(mean.cpu(), x.double().mean(-1)) < 1e-5
    dy = bf(torch.randn(m, d, generator=g(33)))
    ref.backward(dy.double())
    dx0 = torch.randn(m, d, generator=g(34))
    dx = dx0.cuda().clone()
    dg = torch.zeros(d, dtype=torch.float32, device="cuda")
    db = torch.zeros(d, dtype=torch.float32, device="cuda")
    K.layernorm_bwd(dy.cuda(), x.cuda(), d, mean, rstd, gamma.cuda(), dx, d, True, dg, db, m, d)
    assert rel_l2(dx.cpu(), dx0.double() + xd.grad) < 1e-5
    assert rel_l2(dg.cpu(), gd.grad) < 1e-5 and rel_l2(db.cpu(), bd.grad) < 1e-5
    dx2 = torch.full((m, d), 9.0, dtype=torch.float32, device="cuda")
    K.layernorm_bwd(dy.cuda(), x.cuda(), d, mean, rstd, gamma.cuda(), dx2, d, False, dg, db, m, d)
    assert rel_l2(dx2.cpu(), xd.grad) < 1e-5


def test_layernorm_bwd_fused_dropout_tail():
    from chambers_amd import kernels as K
    m, d, rate, key = 300, 192, 0.1, 31337
    x = torch.randn(m, d, generator=g(36))
    gamma = torch.randn(d, generator=g(37))
    dy = bf(torch.randn(m, d, generator=g(38)))
    mean = x.mean(-1).cuda()
    rstd = (1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-6)).cuda()
    dx0 = torch.randn(m, d, generator=g(39))
    dxa, dxb = dx0.cuda().clone(), dx0.cuda().clone()
    dg1, db1, dg2, db2 = (torch.zeros(d, device="cuda") for _ in range(4))
    K.layernorm_bwd(dy.cuda(), x.cuda(), d, mean, rstd, gamma.cuda(), dxa, d, True, dg1, db1, m, d)
    dz = torch.zeros(m, d, dtype=torch.bfloat16, device="cuda")
    zsum = torch.zeros(d, device="cuda")
    K.layernorm_bwd(dy.cuda(), x.cuda(), d, mean, rstd, gamma.cuda(), dxb, d, True, dg2, db2, m, d, dz=dz, dz_colsum=zsum, drop_rate=rate,
                    drop_key=key)
    assert torch.equal(dxa, dxb) and rel_l2(dg1.cpu(), dg2.cpu()) < 1e-5 and rel_l2(db1.cpu(), db2.cpu()) < 1e-5   # atomics: order-dependent last bits
    ref = torch.empty(m, d, dtype=torch.bfloat16, device="cuda")
    K.dropout_bwd(dxa, ref, m, d, rate, key)
    assert torch.equal(dz, ref)
    cs = torch.zeros(d, device="cuda")
    K.colsum(ref, cs)
    assert rel_l2(zsum.cpu(), cs.cpu()) < 1e-5


@pytest.mark.parametrize("rate", [0.0, 0.1])
def test_layernorm_bwd_class_rows_with_the_fused_dropout_tail(rate):
    """The final LayerNorm's backward over the class rows (strided, gaps zero-filled) also emits dz = bf16(dropout-backward of dx)
    for EVERY token row (zeros between the class rows) and its column sums: equal to the separate dropout-backward + colsum passes."""
    from chambers_amd import kernels as K
    bsz, n, d, key = 5, 17, 192, 4242
    x = torch.randn(bsz * n, d, generator=g(91)).cuda()
    gamma = torch.randn(d, generator=g(92)).cuda()
    dy = bf(torch.randn(bsz, d, generator=g(93))).cuda()
    xr = x.view(bsz, n, d)[:, 0, :]
    mean = xr.mean(-1).contiguous()
    rstd = (1.0 / torch.sqrt(xr.var(-1, unbiased=False) + 1e-6)).contiguous()
    dg1, db1, dg2, db2 = (torch.zeros(d, device="cuda") for _ in range(4))
    dx_a = torch.full((bsz * n, d), 3.0, device="cuda")
    K.layernorm_bwd(dy, x, n * d, mean, rstd, gamma, dx_a, n * d, False, dg1, db1, bsz, d, zero_gaps=True)
    ref = torch.empty(bsz * n, d, dtype=torch.bfloat16, device="cuda")
    K.dropout_bwd(dx_a, ref, bsz * n, d, rate, key)
    cs = torch.zeros(d, device="cuda")
    K.colsum(ref, cs)
    dx_b = torch.full((bsz * n, d), 3.0, device="cuda")
    dz = torch.full((bsz * n + 2, d), 9.0, dtype=torch.bfloat16, device="cuda")
    zsum = torch.zeros(d, device="cuda")
    K.layernorm_bwd(dy, x, n * d, mean, rstd, gamma, dx_b, n * d, False, dg2, db2, bsz, d, dz=dz, dz_colsum=zsum, drop_rate=rate, drop_key=key,
                    zero_gaps=True)
    assert torch.equal(dx_a, dx_b) and torch.equal(dz[:bsz * n], ref) and bool((dz[bsz * n:] == 9.0).all())
    assert rel_l2(zsum.cpu(), cs.cpu()) < 1e-5
    with pytest.raises(ValueError):            # strided rows without the gap fill leave the rows between undefined: refused
        K.layernorm_bwd(dy, x, n * d, mean, rstd, gamma, dx_b, n * d, False, dg2, db2, bsz, d, dz=dz, dz_colsum=zsum)


def test_layernorm_strided_rows():
    from chambers_amd import kernels as K
    bsz, n, d = 4, 5, 192
    x = torch.randn(bsz * n, d, generator=g(35)).cuda()
    gamma, beta = torch.ones(d).cuda(), torch.zeros(d).cuda()
    y = torch.empty(bsz, d, dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.empty(bsz, device="cuda"), torch.empty(bsz, device="cuda")
    K.layernorm_fwd(x, n * d, gamma, beta, y, mean, rstd, bsz, d, 1e-6)
    ref = torch.nn.functional.layer_norm(x[::n].double().cpu(), (d,), None, None, 1e-6)
    assert rel_l2(y.float().cpu(), ref) < 3e-3


# ------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, bsz, n, h, rate, key):
    """fp64 restatement on the given (bf16-valued) qkv; returns o, lse and autograd handles."""
    d = h * 64
    t = qkv.double().reshape(bsz, n, 3, h, 64).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    s = (q @ k.transpose(-1, -2)) / 8.0
    p = torch.softmax(s, dim=-1)
    lse = torch.logsumexp(s, dim=-1)
    if rate:
        keep = torch.from_numpy(rng_ref.attn_keep_mask(tuple(p.shape), key, rate))
        p = p * float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate))) * keep
    o = (p @ v).permute(0, 2, 1, 3).reshape(bsz * n, d)
    return o, lse


@pytest.mark.parametrize("bsz,n,h,rate", [(2, 197, 3, 0.0), (2, 197, 3, 0.1), (3, 17, 2, 0.1), (1, 64, 1, 0.0), (2, 33, 4, 0.5),
                                          (1, 128, 2, 0.1), (1, 577, 2, 0.1), (1, 1, 1, 0.0), (2, 224, 1, 0.1), (2, 225, 2, 0.1),
                                          (1, 577, 3, 0.0), (1, 608, 1, 0.1), (2, 257, 1, 0.25), (1, 785, 1, 0.1),
                                          # more heads than CUs: the persistent backward walks two heads per workgroup (2 / 4 / 7 blocks per head)
                                          (40, 50, 8, 0.1), (30, 100, 9, 0.1), (23, 197, 12, 0.1)])
@pytest.mark.parametrize("variant", ["dbias", "lean", "lean_bits"])
def test_attention_fwd_bwd(bsz, n, h, rate, variant):
    """variant: "dbias" = backward with the fused QKV bias gradient (the 16-wave kernel with the bias epilogue, or the two-pass
    kernels); "lean" = the lean one-pass kernel regenerating the dropout mask by hashing; "lean_bits" = the same kernel testing
    the keep bits the forward wrote."""
    from chambers_amd import kernels as K
    d = h * 64
    key = 0x1234567
    bits = K.attention_drop_bits(bsz, n, h) if (variant == "lean_bits" and n <= 224) else None     # no kernel writes bits beyond 224 tokens
    if bits is not None:
        bits.fill_(-1)
    qkv = bf(torch.randn(bsz * n, 3 * d, generator=g(40)))
    qkv_ref = qkv.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qkv_ref, bsz, n, h, rate, key)
    o = torch.empty(bsz * n, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(bsz * h * n, dtype=torch.float32, device="cuda")
    K.attention_fwd(qkv.cuda(), o, lse, bsz, n, h, 64, rate, key, drop_bits=bits)
    scale = float(o_ref.detach().abs().max())
    err = (o.float().cpu().double() - o_ref.detach()).abs().max().item()
    assert err <= scale * 2 ** -7, "attention fwd max err %g vs scale %g" % (err, scale)
    assert rel_l2(o.float().cpu(), o_ref.detach()) < 4e-3      # bf16 storage of o + bf16 P: ~2e-3
    assert rel_l2(lse.cpu().reshape(bsz, h, n), lse_ref.detach()) < 1e-5
    do = bf(torch.randn(bsz * n, d, generator=g(41)))   # n > 224 runs the two-pass backward (dK/dV pass + dQ pass)
    o_ref.backward(do.double())
    dqkv = torch.zeros(bsz * n, 3 * d, dtype=torch.bfloat16, device="cuda")
    gref = qkv_ref.grad
    if variant == "dbias":
        dbias = torch.zeros(3 * d, device="cuda")
        K.attention_bwd(qkv.cuda(), o, do.cuda(), lse, dqkv, bsz, n, h, 64, rate, key, dbias=dbias)
        rb = rel_l2(dbias.cpu()[d:], gref.sum(0)[d:])             # fused bias gradient (column sums); the key part is ~0 by symmetry
        assert rel_l2(dbias.cpu()[:d], gref.sum(0)[:d]) < 1e-2 and rel_l2(dbias.cpu()[2 * d:], gref.sum(0)[2 * d:]) < 1e-2, rb
        assert float(dbias[d:2 * d].abs().max()) < 2e-2 * float(dbias.abs().max() + 1e-6)
    else:
        K.attention_bwd(qkv.cuda(), o, do.cuda(), lse, dqkv, bsz, n, h, 64, rate, key, drop_bits=bits)
        if bits is not None and rate and n <= 224:
            # the saved bits ARE the mask definition: bit 16(r&1) + 8(r>>1) + (t & 7) of word ((bh*N + q)*4 + g)*2 + (t >> 3) <-> key 16t + 4g + r
            keep = rng_ref.attn_keep_mask((bsz, h, n, n), key, rate)
            w = bits.cpu().numpy().view(np.uint32).reshape(bsz, h, n, 4, 2)
            kk = np.arange(n)
            got = (w[..., (kk >> 2) & 3, kk >> 7] >> (16 * (kk & 1) + 8 * ((kk >> 1) & 1) + ((kk >> 4) & 7)).astype(np.uint32)) & 1
            np.testing.assert_array_equal(got.astype(bool), keep)
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        r = rel_l2(dqkv[:, sl].float().cpu(), gref[:, sl])
        assert r < 1e-2, "%s rel-l2 %g" % (name, r)


@pytest.mark.parametrize("bsz,n,h,rate", [(2, 197, 3, 0.0), (2, 197, 3, 0.1), (1, 193, 1, 0.1), (3, 198, 2, 0.1), (2, 208, 2, 0.25), (1, 208, 1, 0.0),
                                          # more heads than CUs: some workgroups of the persistent kernel walk two heads (head boundaries inside the stream of steps)
                                          (23, 197, 12, 0.1), (23, 197, 12, 0.0), (45, 200, 13, 0.1)])
def test_attention_bwd_pipelined_is_bit_identical_to_lean(bsz, n, h, rate):
    """193 <= N <= 208 runs the persistent pipelined backward (attn_bwd_pipe_kernel: one workgroup per CU, Q / dO / O / keep-bit rings
    by LDS-DMA, a producer wave, the next head's K / V prefetched) - the same MFMA products and sums in the same order as the lean
    one-workgroup-per-head kernel (CHB_ATTN_BWD_ALGO = 4), so dQ, dK and dV agree bit for bit."""
    from chambers_amd import _lib, kernels as K
    d = h * 64
    key = 0x2468ace
    qkv = bf(torch.randn(bsz * n, 3 * d, generator=g(46))).cuda()
    do = bf(torch.randn(bsz * n, d, generator=g(47))).cuda()
    o = torch.empty(bsz * n, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(bsz * h * n, dtype=torch.float32, device="cuda")
    bits = K.attention_drop_bits(bsz, n, h) if rate else None
    K.attention_fwd(qkv, o, lse, bsz, n, h, 64, rate, key, drop_bits=bits)
    outs = []
    try:
        for algo in (4, 0):
            _lib.set_option("ATTN_BWD_ALGO", algo)
            dqkv = torch.full((bsz * n, 3 * d), float("nan"), dtype=torch.bfloat16, device="cuda")
            K.attention_bwd(qkv, o, do, lse, dqkv, bsz, n, h, 64, rate, key, drop_bits=bits)
            torch.cuda.synchronize()
            outs.append(dqkv.view(torch.int16).cpu())
    finally:
        _lib.set_option("ATTN_BWD_ALGO", 0)
    assert not torch.isnan(outs[1].view(torch.bfloat16).float()).any()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("bsz,n,h,rate", [(2, 197, 3, 0.0), (2, 197, 3, 0.1), (1, 193, 1, 0.1), (3, 198, 2, 0.1), (2, 208, 2, 0.25), (23, 197, 12, 0.1),
                                          (23, 197, 12, 0.0), (45, 200, 13, 0.1)])
def test_attention_fwd_pipelined_is_bit_identical_to_whole_head(bsz, n, h, rate):
    """193 <= N <= 208 runs the persistent pipelined forward (attn_fwd_pipe_kernel: heads drawn from a device counter, the next head's
    K / V / Q images loaded by three loader waves while 13 waves compute) - the chunk body is the whole-head kernel's
    (CHB_ATTN_FWD_ALGO = 3), so o, lse and the keep bits agree bit for bit; more heads than CUs in the last cases."""
    from chambers_amd import _lib, kernels as K
    d = h * 64
    qkv = bf(torch.randn(bsz * n, 3 * d, generator=g(48)) * 1.3).cuda()
    outs = []
    try:
        for algo in (3, 0):
            _lib.set_option("ATTN_FWD_ALGO", algo)
            o = torch.full((bsz * n, d), float("nan"), dtype=torch.bfloat16, device="cuda")
            lse = torch.full((bsz * h * n,), float("nan"), dtype=torch.float32, device="cuda")
            bits = K.attention_drop_bits(bsz, n, h) if rate else None
            if bits is not None:
                bits.fill_(-1)
            K.attention_fwd(qkv, o, lse, bsz, n, h, 64, rate, 0x1357, drop_bits=bits)
            torch.cuda.synchronize()
            outs.append((o.view(torch.int16).cpu(), lse.view(torch.int32).cpu(), None if bits is None else bits.cpu()))
    finally:
        _lib.set_option("ATTN_FWD_ALGO", 0)
    a, b = outs
    assert not torch.isnan(b[0].view(torch.bfloat16).float()).any()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and (a[2] is None or torch.equal(a[2], b[2]))


def test_persistent_attention_launches_of_two_streams_do_not_share_a_head_counter():
    """The persistent kernels draw heads from a device counter: one slot per (device, stream).  Forward + backward of two different
    problems queued on two streams at once (several rounds, so that launches overlap in every phase) give the bytes each problem
    gives alone."""
    from chambers_amd import kernels as K
    n, h = 197, 12
    d = h * 64
    probs = []
    for bsz, seed in ((40, 50), (27, 60)):
        qkv = bf(torch.randn(bsz * n, 3 * d, generator=g(seed))).cuda()
        do = bf(torch.randn(bsz * n, d, generator=g(seed + 1))).cuda()
        probs.append((bsz, qkv, do))

    def run(bsz, qkv, do, key):
        o = torch.empty(bsz * n, d, dtype=torch.bfloat16, device="cuda")
        lse = torch.empty(bsz * h * n, dtype=torch.float32, device="cuda")
        bits = K.attention_drop_bits(bsz, n, h)
        dqkv = torch.empty(bsz * n, 3 * d, dtype=torch.bfloat16, device="cuda")
        K.attention_fwd(qkv, o, lse, bsz, n, h, 64, 0.1, key, drop_bits=bits)
        K.attention_bwd(qkv, o, do, lse, dqkv, bsz, n, h, 64, 0.1, key, drop_bits=bits)
        return o, lse, dqkv

    alone = [run(*p, key=0x99 + k) for k, p in enumerate(probs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _round in range(6):
        outs = []
        for k, p in enumerate(probs):
            with torch.cuda.stream(streams[k]):
                outs.append(run(*p, key=0x99 + k))
        torch.cuda.synchronize()
        for a, b in zip(alone, outs):
            assert all(torch.equal(x.view(torch.int16) if x.dtype == torch.bfloat16 else x.view(torch.int32),
                                   y.view(torch.int16) if y.dtype == torch.bfloat16 else y.view(torch.int32)) for x, y in zip(a, b))


@pytest.mark.parametrize("bsz,n,h,rate", [(2, 197, 3, 0.1), (1, 224, 2, 0.0), (3, 50, 1, 0.1), (1, 1, 1, 0.0), (2, 130, 2, 0.5)])
def test_attention_bwd_two_pass_matches_one_pass(bsz, n, h, rate, monkeypatch):
    """The long-sequence backward (N > 224) forced onto short inputs agrees with the LDS-resident one: dK / dV accumulate
    the same MFMA products in the same order (bit-equal), dQ sums the key tiles in another order (bf16 rounding apart)."""
    from chambers_amd import kernels as K
    d = h * 64
    key = 0xabcdef
    qkv = bf(torch.randn(bsz * n, 3 * d, generator=g(43))).cuda()
    do = bf(torch.randn(bsz * n, d, generator=g(44))).cuda()
    o = torch.empty(bsz * n, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(bsz * h * n, dtype=torch.float32, device="cuda")
    K.attention_fwd(qkv, o, lse, bsz, n, h, 64, rate, key)
    outs = []
    from chambers_amd import _lib
    try:
        for algo in (1, 2):
            _lib.set_option("ATTN_BWD_ALGO", algo)
            dqkv = torch.zeros(bsz * n, 3 * d, dtype=torch.bfloat16, device="cuda")
            dbias = torch.zeros(3 * d, device="cuda")
            K.attention_bwd(qkv, o, do, lse, dqkv, bsz, n, h, 64, rate, key, dbias=dbias)
            torch.cuda.synchronize()
            outs.append((dqkv.float().cpu(), dbias.cpu()))
    finally:
        _lib.set_option("ATTN_BWD_ALGO", 0)
    (g1, b1), (g2, b2) = outs
    assert torch.equal(g1[:, d:], g2[:, d:])                          # dK, dV
    assert rel_l2(g2[:, :d], g1[:, :d]) < 3e-3                        # dQ: one bf16 rounding of differently ordered fp32 sums
    assert rel_l2(b2[:d], b1[:d]) < 1e-4 and rel_l2(b2[2 * d:], b1[2 * d:]) < 1e-4


@pytest.mark.parametrize("bsz,n,h,rate", [(2, 197, 3, 0.1), (1, 224, 2, 0.0), (3, 50, 1, 0.1), (1, 1, 1, 0.0), (2, 130, 2, 0.5)])
def test_attention_fwd_streaming_matches_resident(bsz, n, h, rate, monkeypatch):
    """The long-sequence forward (online softmax over 64-key chunks) forced onto short inputs agrees with the LDS-resident
    kernel: same dropout mask, log-sum-exp to fp32 rounding, output to the bf16 rounding of the un-normalised probabilities."""
    from chambers_amd import kernels as K
    d = h * 64
    qkv = bf(torch.randn(bsz * n, 3 * d, generator=g(45)) * 1.5).cuda()
    outs = []
    from chambers_amd import _lib
    try:
        for algo in (1, 2):
            _lib.set_option("ATTN_FWD_ALGO", algo)
            o = torch.empty(bsz * n, d, dtype=torch.bfloat16, device="cuda")
            lse = torch.empty(bsz * h * n, dtype=torch.float32, device="cuda")
            K.attention_fwd(qkv, o, lse, bsz, n, h, 64, rate, 0x51ced)
            torch.cuda.synchronize()
            outs.append((o.float().cpu(), lse.cpu()))
    finally:
        _lib.set_option("ATTN_FWD_ALGO", 0)
    (o1, l1), (o2, l2) = outs
    assert torch.allclose(l1, l2, rtol=1e-6, atol=1e-5)
    assert rel_l2(o2, o1) < 4e-3
    assert torch.equal(o1 == 0, o2 == 0) or rate == 0.0   # identical keep mask (an all-dropped row is zero in both)


@pytest.mark.parametrize("bsz,n,h", [(2, 197, 3), (3, 50, 2), (1, 224, 1)])
@pytest.mark.parametrize("algo", [1, 2])
def test_attention_keep_bits_with_a_forced_forward_algo(bsz, n, h, algo):
    """only the whole-head forward writes the keep bits.  With ATTN_FWD_ALGO forced to the resident (1) or streaming (2)
    kernel a call that hands in drop_bits must still come back with the bits of the mask it applied: the backward that tests them
    then equals the backward that re-hashes the mask, bit for bit, and the bits equal the mask definition."""
    from chambers_amd import _lib, kernels as K
    d, rate, key = h * 64, 0.1, 0x2468ace
    qkv = bf(torch.randn(bsz * n, 3 * d, generator=g(47))).cuda()
    do = bf(torch.randn(bsz * n, d, generator=g(48))).cuda()
    o = torch.empty(bsz * n, d, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(bsz * h * n, dtype=torch.float32, device="cuda")
    bits = K.attention_drop_bits(bsz, n, h)
    bits.fill_(0x5a5a5a5a)                      # stale content: a forward that ignored the pointer would leave it
    try:
        _lib.set_option("ATTN_FWD_ALGO", algo)
        K.attention_fwd(qkv, o, lse, bsz, n, h, 64, rate, key, drop_bits=bits)
    finally:
        _lib.set_option("ATTN_FWD_ALGO", 0)
    o0 = torch.empty_like(o)
    lse0 = torch.empty_like(lse)
    K.attention_fwd(qkv, o0, lse0, bsz, n, h, 64, rate, key)           # the default route without bits: same output
    assert torch.equal(o, o0) and torch.equal(lse, lse0)
    keep = rng_ref.attn_keep_mask((bsz, h, n, n), key, rate)
    w = bits.cpu().numpy().view(np.uint32).reshape(bsz, h, n, 4, 2)
    kk = np.arange(n)
    got = (w[..., (kk >> 2) & 3, kk >> 7] >> (16 * (kk & 1) + 8 * ((kk >> 1) & 1) + ((kk >> 4) & 7)).astype(np.uint32)) & 1
    np.testing.assert_array_equal(got.astype(bool), keep)
    d_bits = torch.zeros(bsz * n, 3 * d, dtype=torch.bfloat16, device="cuda")
    d_hash = torch.zeros_like(d_bits)
    K.attention_bwd(qkv, o, do, lse, d_bits, bsz, n, h, 64, rate, key, drop_bits=bits)
    K.attention_bwd(qkv, o, do, lse, d_hash, bsz, n, h, 64, rate, key)
    assert torch.equal(d_bits, d_hash)


def test_attention_keep_bits_refused_beyond_224_tokens():
    from chambers_amd import _lib, kernels as K
    bsz, n, h = 1, 257, 1
    qkv = torch.zeros(bsz * n, 3 * 64, dtype=torch.bfloat16, device="cuda")
    o = torch.empty(bsz * n, 64, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(bsz * h * n, dtype=torch.float32, device="cuda")
    with pytest.raises(ValueError):            # CHB_EUNSUPPORTED
        K.attention_fwd(qkv, o, lse, bsz, n, h, 64, 0.1, 1, drop_bits=K.attention_drop_bits(bsz, n, h))
    K.attention_fwd(qkv, o, lse, bsz, n, h, 64, 0.0, 1, drop_bits=K.attention_drop_bits(bsz, n, h))   # no dropout: nothing to write, accepted


def test_attention_fwd_fp32_probabilities_tolerance():
    """The north-star bound (1e-3 rel) on the quantity the kernel computes in fp32: the lse / softmax."""
    from chambers_amd import kernels as K
    bsz, n, h = 2, 197, 12
    qkv = bf(torch.randn(bsz * n, 3 * h * 64, generator=g(42)) * 2)
    _, lse_ref = _attn_ref(qkv, bsz, n, h, 0.0, 0)
    o = torch.empty(bsz * n, h * 64, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(bsz * h * n, dtype=torch.float32, device="cuda")
    K.attention_fwd(qkv.cuda(), o, lse, bsz, n, h, 64)
    assert rel_l2(lse.cpu().reshape(bsz, h, n), lse_ref) < 1e-3
    assert torch.allclose(lse.cpu().reshape(bsz, h, n).double(), lse_ref, rtol=1e-3, atol=1e-4)


def test_attention_rejects_unsupported():
    from chambers_amd import kernels as K
    z = torch.zeros(4, 3 * 32, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        K.attention_fwd(z, z, torch.zeros(4, device="cuda"), 1, 4, 1, 32)   # head_dim != 64


# ------------------------------------------------------------------------------------ glue
def test_colsum_dropout_bwd_embed_bwd():
    from chambers_amd import kernels as K
    m, n = 1000, 320
    x = bf(torch.randn(m, n, generator=g(50)))
    out = torch.zeros(n, dtype=torch.float32, device="cuda")
    K.colsum(x.cuda(), out)
    assert rel_l2(out.cpu(), x.double().sum(0)) < 1e-5
    dy = torch.randn(m, n, generator=g(51))
    dz = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    K.dropout_bwd(dy.cuda(), dz, m, n, 0.1, 77)
    keep = torch.from_numpy(rng_ref.keep_mask(m * n, 77, 0.1).reshape(m, n))
    ref = bf(dy * float(np.float32(1) / (np.float32(1) - np.float32(0.1))) * keep)
    assert torch.equal(dz.cpu(), ref)
    bsz, nt, d = 3, 5, 64
    dx = torch.randn(bsz, nt, d, generator=g(52))
    keep = torch.from_numpy(rng_ref.keep_mask(dx.numel(), 5, 0.1).reshape(dx.shape))
    dzr = dx * float(np.float32(1) / (np.float32(1) - np.float32(0.1))) * keep
    dpatch = torch.zeros(bsz * (nt - 1), d, dtype=torch.bfloat16, device="cuda")
    dpos = torch.zeros(nt, d, device="cuda")
    dcls = torch.zeros(d, device="cuda")
    K.embed_bwd(dx.cuda().reshape(bsz * nt, d), dpatch, dpos, dcls, bsz, nt, d, 0.1, 5)
    assert torch.equal(dpatch.cpu(), bf(dzr[:, 1:].reshape(-1, d)))
    assert rel_l2(dpos.cpu(), dzr.double().sum(0)) < 1e-6 and rel_l2(dcls.cpu(), dzr[:, 0].double().sum(0)) < 1e-6


def test_softmax_ce():
    from chambers_amd import kernels as K
    bsz, c, ld = 37, 1000, 1024
    logits = torch.zeros(bsz, ld)
    logits[:, :c] = torch.randn(bsz, c, generator=g(60)) * 3
    labels = torch.randint(0, c, (bsz,), generator=g(61))
    lg = logits[:, :c].double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lg, labels, reduction="none")
    ref.mean().backward()
    loss = torch.empty(bsz, device="cuda")
    dl = torch.full((bsz, ld), 5.0, dtype=torch.bfloat16, device="cuda")
    K.softmax_ce(logits.cuda(), labels.to(torch.int32).cuda(), loss, dl, c, 1.0 / bsz)
    assert rel_l2(loss.cpu(), ref.detach()) < 1e-6
    assert rel_l2(dl[:, :c].float().cpu(), lg.grad) < 4e-3 and float(dl[:, c:].abs().max()) == 0.0


def test_cast_transpose_and_adamw():
    from chambers_amd import kernels as K
    from oracle import vit_ref
    shapes = [(100, 36), (64, 64), (7, 300)]
    offs, tot = [], 0
    for r, c in shapes:
        offs.append(tot)
        tot += (r * c + 1023) // 1024 * 1024
    src = torch.randn(tot, generator=g(70))
    desc = torch.tensor([[o, o, r, c] for o, (r, c) in zip(offs, shapes)], dtype=torch.int64)
    dst = torch.zeros(tot, dtype=torch.bfloat16, device="cuda")
    dstt = torch.zeros(tot, dtype=torch.bfloat16, device="cuda")
    K.cast_transpose(src.cuda(), dst, dstt, desc.cuda(), len(shapes), max(((r + 63) // 64) * ((c + 63) // 64) for r, c in shapes))
    for o, (r, c) in zip(offs, shapes):
        m = bf(src[o:o + r * c].reshape(r, c))
        assert torch.equal(dst[o:o + r * c].cpu().reshape(r, c), m)
        assert torch.equal(dstt[o:o + r * c].cpu().reshape(c, r), m.t().contiguous())
    # AdamW, 3 steps, decay on the first 1024-chunk only
    n = 4096
    p0 = torch.randn(n, generator=g(71))
    params = {"a": p0[:1024].clone(), "b": p0[1024:].clone()}
    m_ = {k: torch.zeros_like(v) for k, v in params.items()}
    v_ = {k: torch.zeros_like(v) for k, v in params.items()}
    p = p0.cuda().clone()
    mm, vv = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    flags = torch.tensor([1, 0, 0, 0], dtype=torch.uint8, device="cuda")
    for step in range(1, 4):
        grad = torch.randn(n, generator=g(72 + step))
        vit_ref.adamw_step(params, {"a": grad[:1024], "b": grad[1024:]}, m_, v_, step, lr=1e-3, weight_decay=0.05,
                           decay_mask={"a": True, "b": False})
        b1, b2 = np.float32(0.9), np.float32(0.999)
        lr_t = np.float32(1e-3) * np.sqrt(np.float32(1) - np.power(b2, np.float32(step))) / (np.float32(1) - np.power(b1, np.float32(step)))
        K.adamw(p, grad.cuda(), mm, vv, flags, float(lr_t), 0.9, 0.999, 1e-7, 0.05)
    ref = torch.cat([params["a"], params["b"]])
    assert torch.allclose(p.cpu(), ref, rtol=2e-6, atol=1e-7), float((p.cpu() - ref).abs().max())


def test_tensor_utilities_of_the_standalone_layers():
    """chb_add_f32 / chb_cast_* / chb_copy_rows / chb_softmax_f32: the arithmetic the stand-alone Keras-style layers used to ask torch for."""
    from chambers_amd import kernels as K
    a = torch.randn(1027, 33, generator=g(81)).cuda()
    b = torch.randn(1027, 33, generator=g(82)).cuda()
    assert torch.equal(K.add_f32(a, b), a + b)
    assert torch.equal(K.cast_bf16(a), a.to(torch.bfloat16))
    assert torch.equal(K.cast_f32(K.cast_bf16(a)), a.to(torch.bfloat16).float())
    parts = [torch.randn(5, n, 64, generator=g(83 + n)).cuda() for n in (1, 196, 2)]
    assert torch.equal(K.concat_axis1(parts), torch.cat(parts, dim=1))
    x = (torch.randn(37, 1000, generator=g(90)) * 4).cuda()
    got = K.softmax_rows(x)
    ref = torch.softmax(x.double(), dim=-1)
    assert float((got.double() - ref).abs().max()) < 3.6e-7 and float((got.sum(-1) - 1).abs().max()) < 1e-5      # fp32 expf: measured 2.4e-7
    from chambers_amd.layers.embedding import ConcatEmbedding
    layer = ConcatEmbedding(1, 64, side="left", axis=1)
    tokens = torch.randn(3, 10, 64, generator=g(91)).cuda()
    out = layer(tokens)
    assert out.shape == (3, 11, 64) and torch.equal(out[:, 1:], tokens) and torch.equal(out[:, 0], layer.embedding.value.expand(3, 64))
