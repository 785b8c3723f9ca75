"""Randomness contract of the build (SURVEY §5 'RNG').

The reference draws every random decision from TF's stateful global generator
(augmentations/image_augmentations.py:54,523,608; keras Dropout).  Here every decision is
explicit: augmentation layers take their decisions as arguments or draw them from the host
generator below; dropout sites get a 32-bit key = f(seed, step, site) and the kernels hash
(element index, key) — forward and backward regenerate identical masks, nothing is stored.
"""
import numpy as np

_M64 = (1 << 64) - 1
_host = np.random.Generator(np.random.PCG64(42))


def set_seed(seed):
    """Seed the host generator used for augmentation decisions (cf. utils/generic.py:43-51)."""
    global _host
    _host = np.random.Generator(np.random.PCG64(int(seed)))


def host_generator():
    return _host


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def site_key(seed, step, site):
    """32-bit dropout key of site `site` at optimisation step `step`."""
    z = _splitmix64((int(seed) & _M64) ^ _splitmix64((int(step) << 20) + int(site) + 1))
    return int(z & 0xFFFFFFFF)


# dropout-site numbering of the ViT graph
SITE_EMBED = 0


def site_attn(layer):
    return 1 + 3 * layer


def site_proj(layer):
    return 2 + 3 * layer


def site_mlp(layer):
    return 3 + 3 * layer
