"""Eager vs HIP-graph inference of ViT-B/16 at small batch: python tools/inference_latency.py [batch]."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = ViTConfig(16, 768, 12, 12, 3072, dropout_rate=0.1, image_size=(224, 224), classes=1000)
eng = ViTEngine(cfg, B, training=False)
eng.load_keras_weights(init_keras_weights(cfg, seed=1))
x = torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8, device="cuda")


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = t(lambda: eng.forward(x, training=False))
run = eng.capture_inference()
graph = t(lambda: run(x))
print("ViT-B/16 inference batch %d: eager %.3f ms (%.0f img/s)   HIP graph %.3f ms (%.0f img/s)" % (B, eager, B / eager * 1e3, graph, B / graph * 1e3))
