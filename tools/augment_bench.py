"""Achieved algorithmic GB/s of every augmentation kernel on a [256,224,224,3] uint8 batch (SURVEY §8d figures:
2*H*W*3 bytes per image for single-pass ops, 3*H*W*3 for AutoContrast/Equalize, H*W*3*(1+out_bytes) for normalisation)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chambers_amd import augmentations as aug, kernels as K

B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 224, 224
x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
px = B * H * W * 3
centers = torch.full((B, 2), 100, dtype=torch.int32, device="cuda")   # resident decisions: the H2D copy of a host array is not the kernel
flip_bits = (torch.arange(B, device="cuda") % 2 == 0)   # resident decisions
cases = [
    ("Invert", lambda: aug.Invert()(x), 2), ("Posterize", lambda: aug.Posterize(3)(x), 2), ("Solarize", lambda: aug.Solarize(230)(x), 2),
    ("SolarizeAdd", lambda: aug.SolarizeAdd(99)(x), 2), ("Brightness", lambda: aug.Brightness(1.72)(x), 2),
    ("Contrast", lambda: aug.Contrast(1.72)(x), 2), ("Color", lambda: aug.Color(1.72)(x), 2), ("Sharpness", lambda: aug.Sharpness(1.72)(x), 2),
    ("ShearX", lambda: aug.ShearX(0.27, fill_value=128)(x, negate=False), 2), ("TranslateY", lambda: aug.TranslateY(90.0, fill_value=128)(x, negate=True), 2),
    ("Rotate", lambda: aug.Rotate(27.0, fill_value=128)(x, negate=False), 2), ("CutOut", lambda: aug.CutOut(72, 128)(x, centers=centers), 2),
    ("AutoContrast", lambda: aug.AutoContrast()(x), 3), ("Equalize", lambda: aug.Equalize()(x), 3),
    ("Normalize_tf_f32", lambda: aug.ImageNetNormalization("tf")(x), 5), ("NormalizePatchify_bf16", lambda: K.normalize_patchify(x, 16, "tf"), 3),
    # input side: algorithmic bytes = window read once + output written once, as multiples of the [B,224,224,3] uint8 batch
    ("RandomFlip", lambda: aug.RandomFlip()(x, training=True, flip_horizontal=flip_bits, flip_vertical=flip_bits), 2),
    ("RandomCrop_192", lambda: aug.RandomCrop(192, 192)(x, training=True, offset=(11, 13)), 2 * (192 * 192) / (224.0 * 224.0)),
    ("Rescaling_f32", lambda: aug.Rescaling(1 / 255.0)(x), 5),
    ("Resizing_224_to_256_bilinear_f32", lambda: aug.Resizing(256, 256)(x), 1 + 4 * (256 * 256) / (224.0 * 224.0)),
    ("Resizing_224_to_160_nearest_u8", lambda: aug.Resizing(160, 160, interpolation="nearest")(x), 2 * (160 * 160) / (224.0 * 224.0)),
]
out = {}
REP = 20
for name, fn, mult in cases:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    # replay the REP launches from a HIP graph: the kernels are 10-60 us, shorter than the Python dispatch of one layer call,
    # so back-to-back eager calls would time the host, not the kernel
    mode = "graph"
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(REP):
                fn()
        run = graph.replay
    except Exception as exc:   # noqa: BLE001 - fall back to eager timing, and say so
        mode = "eager (%s)" % type(exc).__name__
        run = lambda: [fn() for _ in range(REP)]   # noqa: E731
    run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    run()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / REP
    out[name] = {"us": round(ms * 1e3, 1), "algorithmic_GBps": round(mult * px / ms / 1e6, 1), "frac_of_8TBps": round(mult * px / ms / 1e6 / 8000, 3),
                 "timing": mode}
    print("%-24s %8.1f us  %8.1f GB/s  (%.1f%% of 8 TB/s)  [%s]" % (name, ms * 1e3, mult * px / ms / 1e6, 100 * mult * px / ms / 1e6 / 8000, mode),
          flush=True)
print(json.dumps({"batch": B, "ops": out}))
