"""Do the dgrad chain (persistent NT GEMMs) and the weight-gradient GEMMs (TN) of one ViT-B/16 block run faster from TWO streams
than back to back on one?  Kernels of one stream are separated by a full barrier, so a launch's tail (4.6 rounds of tiles on the
N = 768 shapes) idles CUs that an independent kernel of another stream could use.  B = 512 shapes, independent buffers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chambers_amd import _lib, kernels as K

M = 512 * 197
dev = "cuda"
bf = torch.bfloat16
nt = [("fc2_dgrad", 3072, 768), ("fc1_dgrad", 768, 3072), ("proj_dgrad", 768, 768), ("qkv_dgrad", 768, 2304)]
tn = [("fc2_wgrad", 3072, 768), ("fc1_wgrad", 768, 3072), ("proj_wgrad", 768, 768), ("qkv_wgrad", 768, 2304)]
A = {k: torch.randn(M, k, device=dev).to(bf) for k in (768, 2304, 3072)}
Bw = {(n, k): torch.randn(n, k, device=dev).to(bf) for _, n, k in nt}
outs = {(n, k): torch.empty(M, n, dtype=bf, device=dev) for _, n, k in nt}
dw = {(kd, nd): torch.zeros(kd, nd, device=dev) for _, kd, nd in tn}
ws = torch.empty(max(K.tn_workspace_elems(kd, nd) for _, kd, nd in tn), device=dev)
LAYERS = 12


def chain_nt():
    for _, n, k in nt:
        K.gemm_nt(A[k], Bw[(n, k)], outs[(n, k)])


def chain_tn():
    for _, kd, nd in tn:
        K.gemm_tn(A[kd], A[nd], dw[(kd, nd)], ws=ws)


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


side = torch.cuda.Stream()


def serial():
    for _ in range(LAYERS):
        chain_nt()
        chain_tn()


def two_streams():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    for _ in range(LAYERS):
        chain_nt()
        with torch.cuda.stream(side):
            chain_tn()
    main.wait_stream(side)


for q in (0, 1):
    _lib.set_option("GEMM_TILE_QUEUE", q)
    a = [timed(serial) for _ in range(3)]
    b = [timed(two_streams) for _ in range(3)]
    print("GEMM_TILE_QUEUE=%d  one stream %.2f ms   two streams %.2f ms   (%+.1f %%)" % (q, min(a), min(b), 100 * (min(b) / min(a) - 1)), flush=True)
_lib.set_option("GEMM_TILE_QUEUE", 0)
