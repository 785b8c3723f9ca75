import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chambers_amd import kernels as K
B, N, H = 512, 197, 12
D = H * 64
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda")
do = torch.randn(B * N, D, device="cuda").to(torch.bfloat16)
dqkv = torch.empty(B * N, 3 * D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * H * N, device="cuda")
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for rate in (0.0, 0.1):
    f = t(lambda: K.attention_fwd(qkv, o, lse, B, N, H, 64, rate, 7))
    b = t(lambda: K.attention_bwd(qkv, o, do, lse, dqkv, B, N, H, 64, rate, 7))
    fl = 4.0 * B * H * N * N * 64
    print("rate %.1f  fwd %.3f ms (%.0f TF/s)   bwd %.3f ms (%.0f TF/s)" % (rate, f, fl / f / 1e9, b, 2.5 * fl / b / 1e9), flush=True)
dbias = torch.zeros(3 * D, device="cuda")
for rep in range(2):
    b0 = t(lambda: K.attention_bwd(qkv, o, do, lse, dqkv, B, N, H, 64, 0.1, 7))
    b1 = t(lambda: K.attention_bwd(qkv, o, do, lse, dqkv, B, N, H, 64, 0.1, 7, dbias=dbias))
    cs = torch.zeros(3 * D, device="cuda")
    b2 = t(lambda: K.colsum(dqkv, cs))
    print("bwd no-bias %.3f ms | fused bias %.3f ms | separate colsum %.3f ms" % (b0, b1, b2))
