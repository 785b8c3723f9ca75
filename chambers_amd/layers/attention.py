"""chambers.layers.attention on MI355X (reference: chambers/layers/attention.py:6-181).

`MultiHeadAttention` keeps the reference's weights — w_query/b_query, w_value/b_value, w_key/b_key,
w_projection/b_projection, in that creation order and with those shapes (:54-96) — and runs
QKV projection (one fused bf16 MFMA GEMM) -> fused attention kernel -> output projection."""
import math

import torch

from .. import kernels as K
from .._keras_like import Layer, register_keras_serializable
from . import autograd as AG
from .core import _next_key


@register_keras_serializable(package="Chambers")
class ScaledAttention(Layer):
    """keras Attention with scores / sqrt(key_dim) (:7-23).  Inputs [query, value, key] are
    [B, heads, T, head_dim] tensors; T and head_dim must match what the fused kernel supports."""

    _site = 9100

    def __init__(self, key_dim=None, causal=False, dropout=0.0, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        if causal:
            raise ValueError("causal attention is not on the ViT path (layers/transformer.py:31-37 builds it with causal=False)")
        self.key_dim = key_dim
        self.causal = causal
        self.dropout = dropout
        self._scale = math.sqrt(key_dim) if key_dim is not None else None

    def call(self, inputs, mask=None, training=None, key=None, **kwargs):
        if mask is not None and any(m is not None for m in mask):
            raise ValueError("attention masks are not on the ViT path")
        q, v = inputs[0], inputs[1]
        k = inputs[2] if len(inputs) > 2 else v
        b, h, t, hd = q.shape
        if k.shape != q.shape or v.shape != q.shape:
            raise ValueError("the fused attention kernel needs equal query/key/value shapes (self-attention)")
        d = h * hd
        # pack [q | k | v] as the fused kernel wants them: [B*T, 3*H*hd] (pure data movement: permute + concatenate; autograd routes
        # the packed gradient back through the same views)
        qkv = torch.cat([x.permute(0, 2, 1, 3).reshape(b * t, d) for x in (q, k, v)], dim=1)
        rate = self.dropout if training else 0.0
        o = AG.AttentionFn.apply(qkv, b, t, h, hd, rate, (_next_key(self._site) if key is None else key) if rate else 0)
        return o.reshape(b, t, h, hd).permute(0, 2, 1, 3)

    def get_config(self):
        return dict(super().get_config(), key_dim=self.key_dim, causal=self.causal, dropout=self.dropout)


@register_keras_serializable(package="Chambers")
class MultiHeadAttention(Layer):
    _site = 9200

    def __init__(self, head_dim=64, num_heads=8, dense_kernel_initializer="glorot_uniform", dropout_rate=0.1, causal=False, **kwargs):
        super(MultiHeadAttention, self).__init__(**kwargs)
        self.head_dim = head_dim
        self.num_heads = num_heads
        self.dense_kernel_initializer = dense_kernel_initializer
        self.dropout_rate = dropout_rate
        self.causal = causal
        self.attention = ScaledAttention(key_dim=head_dim, causal=causal, dropout=dropout_rate)

    def build(self, input_shape):
        d = input_shape[0][-1]
        n, h = self.num_heads, self.head_dim
        init = self.dense_kernel_initializer
        self.w_query = self.add_weight("w_query", (d, n, h), init)
        self.b_query = self.add_weight("b_query", (n, 1, h), "zeros")
        self.w_value = self.add_weight("w_value", (d, n, h), init)
        self.b_value = self.add_weight("b_value", (n, 1, h), "zeros")
        self.w_key = self.add_weight("w_key", (d, n, h), init)
        self.b_key = self.add_weight("b_key", (n, 1, h), "zeros")
        self.w_projection = self.add_weight("w_projection", (n, d, h), init)
        self.b_projection = self.add_weight("b_projection", (1, d), "zeros")

    def _fused_weights(self):
        """[w_query | w_key | w_value] as one [d, 3*n*h] kernel (+ bias) and the projection as [n*h, d]: reshapes and a concatenation
        of the Keras-shaped variables (data movement on the tape, so each variable receives its slice of the fused gradient)."""
        d = self.w_query.shape[0]
        nh = self.num_heads * self.head_dim
        w = torch.cat([self.w_query.value.reshape(d, nh), self.w_key.value.reshape(d, nh), self.w_value.value.reshape(d, nh)], dim=1)
        bqkv = torch.cat([self.b_query.value.reshape(nh), self.b_key.value.reshape(nh), self.b_value.value.reshape(nh)])
        wp = self.w_projection.value.permute(0, 2, 1).reshape(nh, d)                # [(n,h), d]
        return w, bqkv, wp, self.b_projection.value.reshape(d)

    def call(self, inputs, mask=None, training=None, key=None, **kwargs):
        q = inputs[0]
        v = inputs[1]
        k = inputs[2] if len(inputs) > 2 else v
        if not (q is v and q is k):
            raise ValueError("the MI355X path implements self-attention (q is v is k), as EncoderLayer calls it (layers/transformer.py:66-68)")
        if mask is not None and any(m is not None for m in mask):
            raise ValueError("attention masks are not on the ViT path")
        b, t, d = q.shape
        w, bqkv, wp, bp = self._fused_weights()
        qkv = AG.LinearFn.apply(q.reshape(b * t, d), w, bqkv, None, True)                     # bf16 [B*T, 3*n*h]
        rate = self.dropout_rate if training else 0.0
        o = AG.AttentionFn.apply(qkv, b, t, self.num_heads, self.head_dim, rate, (_next_key(self._site) if key is None else key) if rate else 0)
        out = AG.LinearFn.apply(o, wp, bp, None, False)                                        # fp32 [B*T, d]
        return out.reshape(b, t, d)

    def get_config(self):
        config = {"head_dim": self.head_dim, "num_heads": self.num_heads, "dense_kernel_initializer": self.dense_kernel_initializer,
                  "dropout_rate": self.dropout_rate, "causal": self.causal}
        return dict(list(super(MultiHeadAttention, self).get_config().items()) + list(config.items()))
