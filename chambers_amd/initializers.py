"""Host-side weight initialisers matching the ones the reference names
(glorot_uniform: layers/attention.py:32, layers/transformer.py:14; TruncatedNormal(0.02):
vision_transformer.py:254,258; zeros/ones for biases and LayerNormalization).
Initialisation is not on the hot path: NumPy on the host, seeded by `set_seed`."""
import numpy as np

_rng = np.random.Generator(np.random.PCG64(1234))


def set_seed(seed):
    global _rng
    _rng = np.random.Generator(np.random.PCG64(int(seed)))


def _fans(shape):
    """keras `_compute_fans`."""
    if len(shape) < 1:
        return 1, 1
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    receptive = int(np.prod(shape[:-2]))
    return shape[-2] * receptive, shape[-1] * receptive


def glorot_uniform(shape):
    fan_in, fan_out = _fans(shape)
    limit = np.sqrt(6.0 / max(1.0, (fan_in + fan_out)))
    return _rng.uniform(-limit, limit, size=shape).astype(np.float32)


def zeros(shape):
    return np.zeros(shape, dtype=np.float32)


def ones(shape):
    return np.ones(shape, dtype=np.float32)


class TruncatedNormal:
    """tf.keras.initializers.TruncatedNormal: resample beyond two standard deviations."""

    def __init__(self, mean=0.0, stddev=0.05):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape):
        out = _rng.normal(0.0, 1.0, size=shape)
        bad = np.abs(out) > 2.0
        while bad.any():
            out[bad] = _rng.normal(0.0, 1.0, size=int(bad.sum()))
            bad = np.abs(out) > 2.0
        return (self.mean + self.stddev * out).astype(np.float32)

    def get_config(self):
        return {"mean": self.mean, "stddev": self.stddev}


_BY_NAME = {"glorot_uniform": glorot_uniform, "zeros": zeros, "ones": ones}


def get(init):
    if init is None:
        return glorot_uniform  # keras add_weight default for floats
    if callable(init):
        return init
    if isinstance(init, str):
        if init not in _BY_NAME:
            raise ValueError("Unknown initializer: " + init)
        return _BY_NAME[init]
    raise ValueError("Could not interpret initializer identifier: " + repr(init))


def serialize(init):
    if isinstance(init, TruncatedNormal):
        return {"class_name": "TruncatedNormal", "config": init.get_config()}
    return init
