"""Attention kernel timings: python tools/attn_bench.py [B N H].  Forward: persistent pipelined (default at 193..208 tokens), whole-head,
resident, streaming; backward: persistent pipelined (default at 193..208 tokens), lean one-pass with / without the forward's keep bits,
16-wave and 8-wave resident, two-pass, fused bias gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chambers_amd import _lib, kernels as K

B, N, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 197, 12)
D = H * 64
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda")
do = torch.randn(B * N, D, device="cuda").to(torch.bfloat16)
dqkv = torch.empty(B * N, 3 * D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * H * N, device="cuda")
bits = K.attention_drop_bits(B, N, H) if N <= 224 else None   # the keep-bit record exists for heads of <= 224 tokens


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


fl = 4.0 * B * H * N * N * 64
hbm_f = (B * N * 4 * D * 2) / 1e9                     # qkv read + o written, GB
hbm_b = (B * N * (3 * D + 2 * D + 3 * D) * 2) / 1e9   # qkv, o, do read + dqkv written
for rate in (0.0, 0.1):
    line = "B %d N %d H %d rate %.1f\n" % (B, N, H, rate)
    for algo, label in ((0, "pipe (persistent, default for 193..208 tokens)"), (3, "head"), (1, "resident"), (2, "stream")):
        if (algo == 1 and N > 224):
            continue
        _lib.set_option("ATTN_FWD_ALGO", algo)
        f = t(lambda: K.attention_fwd(qkv, o, lse, B, N, H, 64, rate, 7, drop_bits=bits if (rate and bits is not None) else None))
        line += "   fwd[%s] %.3f ms (%.0f TF/s, %.2f TB/s algorithmic)\n" % (label, f, fl / f / 1e9, hbm_f / f)
    _lib.set_option("ATTN_FWD_ALGO", 0)
    K.attention_fwd(qkv, o, lse, B, N, H, 64, rate, 7, drop_bits=bits if (rate and bits is not None) else None)
    for algo, label, kw in ((0, "pipe (persistent, default for 193..208 tokens) + keep bits", dict(drop_bits=bits if (rate and bits is not None) else None)),
                            (4, "lean + keep bits", dict(drop_bits=bits if (rate and bits is not None) else None)), (4, "lean, hashing", {}),
                            (3, "resident 16 waves", {}), (1, "resident 8 waves", {}), (2, "two-pass", {})):
        if algo in (1, 3) and N > 224:
            continue
        _lib.set_option("ATTN_BWD_ALGO", algo)
        b = t(lambda: K.attention_bwd(qkv, o, do, lse, dqkv, B, N, H, 64, rate, 7, **kw))
        line += "   bwd[%s] %.3f ms (%.0f TF/s, %.2f TB/s algorithmic)\n" % (label, b, 2.5 * fl / b / 1e9, hbm_b / b)
    _lib.set_option("ATTN_BWD_ALGO", 0)
    dbias = torch.zeros(3 * D, device="cuda")
    ws = torch.empty(B * 3 * D, device="cuda")
    b = t(lambda: K.attention_bwd(qkv, o, do, lse, dqkv, B, N, H, 64, rate, 7, dbias=dbias, dbias_ws=ws))
    c = t(lambda: K.colsum(dqkv, dbias, m=B * N))
    line += "   bwd[16 waves + fused qkv bias gradient] %.3f ms   (stand-alone column sums of dqkv: %.3f ms)" % (b, c)
    print(line, flush=True)
