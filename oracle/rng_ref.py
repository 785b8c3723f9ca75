"""CPU restatement of the counter-hash RNG the HIP dropout sites use
(TEST INFRASTRUCTURE ONLY — see oracle/augment_ref.py header for the rules).

The reference draws dropout masks from TF's stateful RNG (tf.nn.dropout via
keras Dropout, layers/transformer.py:38,48; attention dropout inside keras
Attention, layers/attention.py:44-46), which is not reproducible outside TF, so
the build defines its own counter-based generator and parity is defined *given
the mask*.  This file is the mask definition the kernels in
chambers_amd/csrc/common.hpp (`chb_hash32`, `chb_keep`) must reproduce bit for
bit; tests compare a mask dumped by the HIP side against it.
"""
import numpy as np

_M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def site_key(seed, step, site):
    """32-bit key of one dropout site for one step (host side, mirrors
    chambers_amd.rng.site_key)."""
    z = splitmix64((int(seed) & _M64) ^ splitmix64((int(step) << 20) + int(site) + 1))
    return int(z & 0xFFFFFFFF)


def hash32(x):
    """'lowbias32' integer finaliser on uint32 arrays."""
    x = np.asarray(x, dtype=np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def threshold(rate):
    return int(round(float(rate) * 65536.0))


def keep_mask(n_elements, key, rate):
    """bool[n]: element e is kept iff u16(e) >= round(rate*65536), where
    u16(e) = 16-bit half (e & 1) of hash32((e >> 1) ^ key)."""
    e = np.arange(n_elements, dtype=np.uint64)
    r = hash32(((e >> np.uint64(1)) ^ np.uint64(key)) & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    u16 = (r >> (np.uint64(16) * (e & np.uint64(1)))) & np.uint64(0xFFFF)
    return u16 >= np.uint64(threshold(rate))


def attn_keep_mask(shape, key, rate):
    """Keep mask of the dropout on the attention probabilities, bool of ``shape`` = [B, H, N, N].  The element index of
    (b, h, q, k) is ((b*H + h)*N + q)*Np4 + k with the ROW STRIDE Np4 = N rounded up to a multiple of 4 (every query row starts
    on a hash-pair boundary, so the four consecutive keys a lane of the kernels holds are the halves of exactly two hashes);
    otherwise as `keep_mask`."""
    b, h, n, n2 = (int(v) for v in shape)
    assert n == n2
    np4 = (n + 3) & ~3
    full = keep_mask(b * h * n * np4, key, rate).reshape(b, h, n, np4)
    return np.ascontiguousarray(full[..., :n])
