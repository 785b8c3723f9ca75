"""Reference ceiling only (NOT used by the product): what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on the same shapes."""
import torch
M = 512 * 197
for name, m, n, k, nt in [("qkv_fwd", M, 2304, 768, True), ("fc1_fwd", M, 3072, 768, True), ("fc2_fwd", M, 768, 3072, True), ("proj_fwd", M, 768, 768, True),
                          ("fc1_wgrad", 768, 3072, M, False), ("fc2_wgrad", 3072, 768, M, False)]:
    if nt:
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16); b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
        f = lambda: torch.matmul(a, b.t())
    else:
        a = torch.randn(k, m, device="cuda").to(torch.bfloat16); b = torch.randn(k, n, device="cuda").to(torch.bfloat16)
        f = lambda: torch.matmul(a.t(), b)
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print("%-10s %.3f ms %.1f TF/s" % (name, ms, 2.0 * m * n * k / ms / 1e9), flush=True)
