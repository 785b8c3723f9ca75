"""Keras core layers the chambers ViT graph is assembled from (tf.keras.layers.{Dense, LayerNormalization,
Dropout, Conv2D, Reshape} as constructed in chambers/layers/transformer.py:38-49 and
chambers/models/backbones/vision_transformer.py:235-283), executing through the HIP C ABI.

Numerics follow the build's mixed-bf16 mode: fp32 variables, bf16 MFMA operands, fp32 accumulation.  Layer
outputs are float32 torch tensors unless stated; `training=True` dropout uses a key derived from
chambers_amd.rng (seed, call counter, layer id)."""
import numpy as np
import torch

from .. import kernels as K
from .. import rng
from .._keras_like import Layer, register_keras_serializable
from ..activations import gelu as _gelu_fn

_call_counter = [0]


def _next_key(site):
    _call_counter[0] += 1
    return rng.site_key(0x5EED, _call_counter[0], site)


def _as2d(x):
    return x.reshape(-1, x.shape[-1])


def _bf16(x):
    return K.cast_bf16(x) if x.dtype in (torch.bfloat16, torch.float32) else x.to(torch.bfloat16)


class _OperandCache:
    """bf16 [N][K] image of an fp32 [K][N] kernel, rebuilt when the owning layer's weights change."""

    def __init__(self):
        self.version, self.wt = -1, None

    def get(self, layer, var):
        if self.version != layer._version or self.wt is None or self.wt.device != var.value.device:
            self.wt = var.value.to(torch.bfloat16).t().contiguous()
            self.version = layer._version
        return self.wt


@register_keras_serializable(package="Chambers")
class Dense(Layer):
    def __init__(self, units, activation=None, kernel_initializer="glorot_uniform", name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.units, self.activation, self.kernel_initializer = int(units), activation, kernel_initializer
        self._cache = _OperandCache()

    def build(self, input_shape):
        self.kernel = self.add_weight("kernel", (input_shape[-1], self.units), self.kernel_initializer)
        self.bias = self.add_weight("bias", (self.units,), "zeros")

    def call(self, inputs, **kwargs):
        lead = inputs.shape[:-1]
        a = _bf16(_as2d(inputs)).contiguous()
        if a.shape[1] % 64 or self.units % 4:
            raise ValueError("Dense on MI355X needs in_features % 64 == 0 and units % 4 == 0 (got %d -> %d)" % (a.shape[1], self.units))
        out = torch.empty((a.shape[0], self.units), dtype=torch.float32, device=a.device)
        act = self.activation
        if act is _gelu_fn or act == "gelu":
            aux = torch.empty((a.shape[0], self.units), dtype=torch.bfloat16, device=a.device)
            K.gemm_nt(a, self._cache.get(self, self.kernel), out, bias=self.bias.value, epilogue=K.EPI_GELU, aux=aux)
        else:
            K.gemm_nt(a, self._cache.get(self, self.kernel), out, bias=self.bias.value)
            if act == "tanh":
                K.tanh_fwd(out)                # `feature` head (vision_transformer.py:275-278)
            elif act == "softmax":
                out = K.softmax_rows(out)          # classifier_activation of the stand-alone layer; training consumes logits
            elif act not in (None, "linear"):
                raise ValueError("unsupported activation %r" % (act,))
        return out.reshape(*lead, self.units)

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[:-1]) + (self.units,)

    def get_config(self):
        act = self.activation
        act = "gelu" if act is _gelu_fn else act
        return dict(super().get_config(), units=self.units, activation=act, kernel_initializer=self.kernel_initializer)


@register_keras_serializable(package="Chambers")
class LayerNormalization(Layer):
    def __init__(self, epsilon=1e-3, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.epsilon = epsilon

    def build(self, input_shape):
        self.gamma = self.add_weight("gamma", (input_shape[-1],), "ones")
        self.beta = self.add_weight("beta", (input_shape[-1],), "zeros")

    def call(self, inputs, **kwargs):
        x = _as2d(inputs).to(torch.float32).contiguous()
        m, d = x.shape
        y = torch.empty((m, d), dtype=torch.bfloat16, device=x.device)
        mean = torch.empty(m, dtype=torch.float32, device=x.device)
        rstd = torch.empty(m, dtype=torch.float32, device=x.device)
        K.layernorm_fwd(x, d, self.gamma.value, self.beta.value, y, mean, rstd, m, d, self.epsilon)
        return y.reshape(inputs.shape)      # bf16: the compute dtype under the mixed policy

    def get_config(self):
        return dict(super().get_config(), epsilon=self.epsilon)


@register_keras_serializable(package="Chambers")
class Dropout(Layer):
    _site = 9000

    def __init__(self, rate, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.rate = rate

    def call(self, inputs, training=None, key=None, **kwargs):
        if not training or self.rate == 0.0:
            return inputs
        key = _next_key(self._site) if key is None else key
        keep = K.dropout_mask(inputs.numel(), self.rate, key, device=inputs.device).reshape(inputs.shape)
        scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(self.rate)))
        return inputs * scale * keep.to(inputs.dtype)

    def get_config(self):
        return dict(super().get_config(), rate=self.rate)


@register_keras_serializable(package="Chambers")
class Conv2D(Layer):
    """Only the form the reference uses: kernel_size == strides, padding 'valid' (patch embedding)."""

    def __init__(self, filters, kernel_size, strides, padding="valid", name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        if kernel_size != strides or padding != "valid":
            raise ValueError("the MI355X patch-embedding path implements Conv2D(kernel_size == strides, padding='valid')")
        self.filters, self.kernel_size, self.strides, self.padding = int(filters), int(kernel_size), int(strides), padding
        self._cache = _OperandCache()

    def build(self, input_shape):
        p = self.kernel_size
        self.kernel = self.add_weight("kernel", (p, p, input_shape[-1], self.filters), "glorot_uniform")
        self.bias = self.add_weight("bias", (self.filters,), "zeros")

    def _wt(self):
        c = self._cache
        if c.version != self._version or c.wt is None:
            k = self.kernel.value
            c.wt = k.reshape(-1, self.filters).to(torch.bfloat16).t().contiguous()
            c.version = self._version
        return c.wt

    def call(self, inputs, **kwargs):
        p = self.kernel_size
        b, h, w, _ = inputs.shape
        patches = K.normalize_patchify(inputs, p, "tf") if inputs.dtype == torch.uint8 else K.patchify_f32(inputs.to(torch.float32), p)
        out = torch.empty((patches.shape[0], self.filters), dtype=torch.float32, device=inputs.device)
        K.gemm_nt(patches, self._wt(), out, bias=self.bias.value)
        return out.reshape(b, h // p, w // p, self.filters)

    def compute_output_shape(self, input_shape):
        p = self.kernel_size
        return (input_shape[0], input_shape[1] // p, input_shape[2] // p, self.filters)

    def get_config(self):
        return dict(super().get_config(), filters=self.filters, kernel_size=self.kernel_size, strides=self.strides, padding=self.padding)


@register_keras_serializable(package="Chambers")
class Reshape(Layer):
    def __init__(self, target_shape, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.target_shape = list(target_shape)

    def call(self, inputs, **kwargs):
        return inputs.reshape(inputs.shape[0], *self.target_shape)

    def get_config(self):
        return dict(super().get_config(), target_shape=self.target_shape)
