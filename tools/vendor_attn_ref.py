"""Reference ceiling only (NOT used by the product): torch's scaled_dot_product_attention (the ROCm flash / CK / aotriton backend it picks)
on the ViT-B/16 attention shape, forward and forward + backward, bf16, with and without dropout."""
import torch

B, H, N, D = 512, 12, 197, 64
q, k, v = (torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
do = torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16)


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for rate in (0.0, 0.1):
    def fwd():
        with torch.no_grad():
            return torch.nn.functional.scaled_dot_product_attention(q, k, v, dropout_p=rate)

    def fwd_bwd():
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v, dropout_p=rate)
        o.backward(do)
        q.grad = k.grad = v.grad = None

    tf, tfb = timed(fwd), timed(fwd_bwd)
    print("dropout %.1f: forward %.3f ms   forward + backward %.3f ms   (backward ~ %.3f ms)" % (rate, tf, tfb, tfb - tf), flush=True)
