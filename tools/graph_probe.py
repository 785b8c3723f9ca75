"""Would a HIP graph of the train step recover the ~6 us between its ~280 dependent launches?  Timing probe only: the captured step
replays with FROZEN dropout keys and Adam scalars (kernel arguments that change every step in the real loop), so its numbers are
not a training run - it answers whether replay shortens the gaps at all."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights

cfg = ViTConfig(patch_size=16, patch_dim=768, n_encoder_layers=12, n_heads=12, ff_dim=3072, dropout_rate=0.1, image_size=(224, 224), classes=1000)
B = 512
eng = ViTEngine(cfg, B, training=True, seed=0)
eng.load_keras_weights(init_keras_weights(cfg, seed=1234))
g = np.random.Generator(np.random.PCG64(0))
images = torch.as_tensor(g.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8), device="cuda")
labels = torch.as_tensor(g.integers(0, 1000, size=(B,)).astype(np.int32), device="cuda")


def step():
    eng.train_step(images, labels, learning_rate=1e-3, weight_decay=0.05)


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(3):
    step()
eager = timed(step)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(graph):
        step()
    replay = timed(graph.replay)
    print("eager %.2f ms/step   graph replay %.2f ms/step   (%+.2f ms)" % (eager, replay, replay - eager))
except Exception as exc:      # noqa: BLE001
    print("eager %.2f ms/step   capture failed: %s: %s" % (eager, type(exc).__name__, str(exc)[:300]))
