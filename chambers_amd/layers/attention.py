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


def _split_mask(mask, b, tq, tk, device):
    """`mask` = [query_mask, value_mask] (either may be None), boolean [batch, length] each (layers/attention.py:129-145: the same
    mask for every head) -> uint8 device tensors or None."""
    if mask is None:
        return None, None
    if not isinstance(mask, (list, tuple)) or len(mask) != 2:
        raise ValueError("mask must be a list [query_mask, value_mask]")
    return K._mask_u8(mask[0], b, tq, device), K._mask_u8(mask[1], b, tk, device)


@register_keras_serializable(package="Chambers")
class ScaledAttention(Layer):
    """keras Attention with scores / sqrt(key_dim) (:7-23).  Inputs [query, value, key] are [B, heads, T, head_dim] tensors.
    Unmasked self-shaped attention with head_dim 64 runs the fused MFMA kernel; `mask=[query_mask, value_mask]`, `causal=True`,
    key / value sequences of another length and other head widths run the general kernel (keras Attention's full semantics)."""

    _site = 9100

    def __init__(self, key_dim=None, causal=False, dropout=0.0, name=None, **kwargs):
        super().__init__(name=name, **kwargs)
        self.key_dim = key_dim
        self.causal = causal
        self.dropout = dropout
        self._scale = math.sqrt(key_dim) if key_dim is not None else None

    def call(self, inputs, mask=None, training=None, key=None, **kwargs):
        q, v = inputs[0], inputs[1]
        k = inputs[2] if len(inputs) > 2 else v
        b, h, t, hd = q.shape
        tk = k.shape[2]
        # scores / sqrt(key_dim) whenever a key_dim was given, whatever the width of the tensors (:13-22); the fused kernel has
        # 1 / sqrt(64) built in, so another scale takes the general kernel
        other_scale = self.key_dim is not None and int(self.key_dim) != int(hd)
        d = h * hd
        rate = self.dropout if training else 0.0
        dkey = (_next_key(self._site) if key is None else key) if rate else 0
        qmask, vmask = _split_mask(mask, b, t, tk, q.device)
        general = qmask is not None or vmask is not None or self.causal or k.shape != q.shape or v.shape != q.shape or hd != 64 or other_scale
        if general:
            # masks / causal / cross-attention / other head widths: keras Attention's full semantics through the general kernel
            flat = lambda x: x.permute(0, 2, 1, 3).reshape(x.shape[0] * x.shape[2], d)     # noqa: E731
            o = AG.AttentionGeneralFn.apply(flat(q), flat(k), flat(v), b, t, tk, h, hd, vmask, qmask, self.causal, rate, dkey,
                                            1.0 / self._scale if other_scale else 0.0)
            return o.reshape(b, t, h, hd).permute(0, 2, 1, 3)
        # pack [q | k | v] as the fused kernel wants them: [B*T, 3*H*hd] (pure data movement: permute + concatenate; autograd routes
        # the packed gradient back through the same views)
        qkv = torch.cat([x.permute(0, 2, 1, 3).reshape(b * t, d) for x in (q, k, v)], dim=1)
        o = AG.AttentionFn.apply(qkv, b, t, h, hd, rate, dkey)
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
        b, t, d = q.shape
        tk = k.shape[1]
        if v.shape[1] != tk:
            raise ValueError("key and value sequences must have the same length, got %d and %d" % (tk, v.shape[1]))
        rate = self.dropout_rate if training else 0.0
        dkey = (_next_key(self._site) if key is None else key) if rate else 0
        qmask, vmask = _split_mask(mask, b, t, tk, q.device)
        w, bqkv, wp, bp = self._fused_weights()
        nh = self.num_heads * self.head_dim
        self_attention = (q is v and q is k)
        if self_attention and qmask is None and vmask is None and not self.causal and self.head_dim == 64:
            qkv = AG.LinearFn.apply(q.reshape(b * t, d), w, bqkv, None, True)                 # bf16 [B*T, 3*n*h]: one fused projection
            o = AG.AttentionFn.apply(qkv, b, t, self.num_heads, self.head_dim, rate, dkey)
        else:
            # layers/attention.py:108-122 in full: separate projections of the three inputs (cross-attention), masks broadcast over the
            # heads (separate_heads_mask), causal mask - through the general attention kernel
            query = AG.LinearFn.apply(q.reshape(b * t, d), w[:, :nh], bqkv[:nh], None, True)
            keyp = AG.LinearFn.apply(k.reshape(b * tk, k.shape[-1]), w[:, nh:2 * nh], bqkv[nh:2 * nh], None, True)
            value = AG.LinearFn.apply(v.reshape(b * tk, v.shape[-1]), w[:, 2 * nh:], bqkv[2 * nh:], None, True)
            o = AG.AttentionGeneralFn.apply(query, keyp, value, b, t, tk, self.num_heads, self.head_dim, vmask, qmask, self.causal, rate, dkey)
        out = AG.LinearFn.apply(o, wp, bp, None, False)                                        # fp32 [B*T, d]
        return out.reshape(b, t, d)

    def compute_mask(self, inputs, mask=None):
        """layers/attention.py:147-153: the query mask passes through."""
        if mask:
            return mask[0]
        return None

    def get_config(self):
        config = {"head_dim": self.head_dim, "num_heads": self.num_heads, "dense_kernel_initializer": self.dense_kernel_initializer,
                  "dropout_rate": self.dropout_rate, "causal": self.causal}
        return dict(list(super(MultiHeadAttention, self).get_config().items()) + list(config.items()))
