"""Single-GPU rehearsal of what a collective running beside the backward pass costs the train step.

An RCCL all-reduce is a kernel of a few dozen workgroups that holds its CUs for the length of the exchange.  This tool puts a
stand-in (tools/cu_hog.hip: n_wg workgroups x 256 threads with 16 KiB of LDS each that stream a 21 MiB buffer, read + write,
for a given time) on a side stream at
every point where the engine hands a gradient bucket to the reducer, exactly as ProcessGroupNCCL orders its kernel behind the
compute stream, and reports the step time with and without it.  Usage:
    python tools/rccl_contention.py [batch] [n_wg] [micros]
"""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_wg = int(sys.argv[2]) if len(sys.argv) > 2 else 32
micros = int(sys.argv[3]) if len(sys.argv) > 3 else 300

so = os.path.join(ROOT, "build", "libcu_hog.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ROOT, "tools", "cu_hog.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tools", "cu_hog.hip")])
hog = ctypes.CDLL(so)
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]

from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights

cfg = ViTConfig(patch_size=16, patch_dim=768, n_encoder_layers=12, n_heads=12, ff_dim=3072, dropout_rate=0.1, image_size=(224, 224), classes=1000)
eng = ViTEngine(cfg, batch, training=True, seed=0)
eng.load_keras_weights(init_keras_weights(cfg, seed=1234))
g = np.random.Generator(np.random.PCG64(0))
images = torch.as_tensor(g.integers(0, 256, size=(batch, 224, 224, 3), dtype=np.uint8), device="cuda")
labels = torch.as_tensor(g.integers(0, 1000, size=(batch,)), device="cuda")
side = torch.cuda.Stream()
hog_buf = torch.zeros(21 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")     # one MLP-side gradient bucket
mode = {"on": False, "placement": "engine"}
launched = [0]


queued = []


def launch():
    if queued:                        # adjacent buckets go out as ONE collective (GradBucketReducer.flush)
        ev = torch.cuda.Event()
        ev.record()                   # the collective starts once the bucket's gradients are final ...
        side.wait_event(ev)
        hog.hog_launch(n_wg, micros, ctypes.c_void_p(hog_buf.data_ptr()), hog_buf.numel() * 4, ctypes.c_void_p(side.cuda_stream))
        launched[0] += 1
    del queued[:]


def bucket_ready(k):
    if mode["on"]:
        queued.append(k)
        if mode["placement"] == "immediate" and k % 2 == 0:     # a block's (or the head's) gradients are complete
            launch()


def flush():                          # the engine's flush points: before each attention backward, end of backward
    if mode["placement"] == "engine":
        launch()


def finish():
    launch()
    if mode["on"]:
        torch.cuda.current_stream().wait_stream(side)     # ... and the optimizer waits for all of them


eng.reducer.bucket_ready = bucket_ready
eng.reducer.flush = flush
eng.reducer.finish = finish
eng.reducer.active = True            # the engine splits a block's backward around the flush point only for an active reducer


def run(steps):
    for _ in range(3):
        eng.train_step(images, labels, learning_rate=1e-3, weight_decay=0.05)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(images, labels, learning_rate=1e-3, weight_decay=0.05)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


from chambers_amd import _lib

print("batch %d  stand-in collective: %d WGs x %d us per gradient bucket (16 KiB LDS each, streaming a 21 MiB buffer)" % (batch, n_wg, micros))
for queue, attn in ((0, 0), (0, 4), (1, 0), (1, 4)):   # static tile shares vs the per-XCD tile queue of the persistent NT GEMM;
    _lib.set_option("GEMM_TILE_QUEUE", queue)           # attention backward: persistent pipelined kernel (heads drawn from a counter) vs
    _lib.set_option("ATTN_BWD_ALGO", attn)              # the lean kernel (one short workgroup per head)
    mode["on"] = False
    base = run(10)
    mode["on"] = True
    print(" GEMM_TILE_QUEUE=%d ATTN_BWD_ALGO=%d   step alone %.2f ms" % (queue, attn, base))
    for placement in ("immediate", "engine"):      # immediate: at bucket_ready (behind a persistent GEMM); engine: where the engine flushes
        mode["placement"] = placement
        eng.dp_flush = "attention" if placement == "engine" else "block"
        launched[0] = 0
        t = run(10)
        print("   placement %-9s %2.0f launches/step   step %.2f ms   (+%.2f ms, %.1f %%)" % (placement, launched[0] / 13, t, t - base, 100 * (t / base - 1)))
