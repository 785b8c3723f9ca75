"""chambers.layers.transformer on MI355X: EncoderLayer / Encoder (reference chambers/layers/transformer.py:8-77,
256-314).  DecoderLayer / Decoder are not on the ViT path and are not built.

The standalone layers execute the same HIP kernels as the whole-model engine, one call per op; the residual
stream is float32 [B, T, D]."""
import torch

from .._keras_like import Layer, register_keras_serializable
from ..activations import gelu
from . import autograd as AG
from .attention import MultiHeadAttention
from .core import Dense, Dropout, LayerNormalization, _next_key


@register_keras_serializable(package="Chambers")
class EncoderLayer(Layer):
    _site = 9300

    def __init__(self, embed_dim=512, num_heads=8, ff_dim=2048, dense_kernel_initializer="glorot_uniform", attention_dropout_rate=0.1,
                 dense_dropout_rate=0.1, norm_epsilon=1e-6, pre_norm=False, **kwargs):
        super(EncoderLayer, self).__init__(**kwargs)
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.ff_dim = ff_dim
        self.dense_kernel_initializer = dense_kernel_initializer
        self.attention_dropout_rate = attention_dropout_rate
        self.dense_dropout_rate = dense_dropout_rate
        self.norm_epsilon = norm_epsilon
        self.pre_norm = pre_norm
        self.multi_head_attention = MultiHeadAttention(head_dim=embed_dim // num_heads, num_heads=num_heads,
                                                       dense_kernel_initializer=dense_kernel_initializer,
                                                       dropout_rate=attention_dropout_rate, causal=False)
        self.dropout1 = Dropout(dense_dropout_rate)
        self.norm1 = LayerNormalization(epsilon=norm_epsilon)
        self.dense1 = Dense(ff_dim, activation=gelu, kernel_initializer=dense_kernel_initializer)
        self.dense2 = Dense(embed_dim, kernel_initializer=dense_kernel_initializer)
        self.dropout2 = Dropout(dense_dropout_rate)
        self.norm2 = LayerNormalization(epsilon=norm_epsilon)
        self.supports_masking = True

    def _sublayers(self):
        # weight order = creation order of the reference: MHA (8), norm1 (2), dense1 (2), dense2 (2), norm2 (2)
        return [self.multi_head_attention, self.norm1, self.dense1, self.dense2, self.norm2]

    def build(self, input_shape):
        shape = tuple(input_shape)
        self.multi_head_attention.build([shape, shape, shape]); self.multi_head_attention.built = True
        self.norm1.build(shape); self.norm1.built = True
        self.dense1.build(shape); self.dense1.built = True
        self.dense2.build(shape[:-1] + (self.ff_dim,)); self.dense2.built = True
        self.norm2.build(shape); self.norm2.built = True

    def _attention_branch(self, h, resid, training, keys, rate):
        """resid + dropout1(multi_head_attention(h)) (layers/transformer.py:53-58,66-69): QKV projection, fused attention, and the output
        projection with dropout + residual in its GEMM epilogue - the engine's fusion, on the autograd tape."""
        m = self.multi_head_attention
        b, t, d = resid.shape
        w, bqkv, wp, bp = m._fused_weights()
        qkv = AG.LinearFn.apply(h.reshape(b * t, d), w, bqkv, None, True)
        arate = m.dropout_rate if training else 0.0
        o = AG.AttentionFn.apply(qkv, b, t, m.num_heads, m.head_dim, arate, (keys.get("attn") or _next_key(m._site)) if arate else 0)
        return AG.LinearResidualFn.apply(o, wp, bp, resid.reshape(b * t, d), rate, (keys.get("proj") or _next_key(self._site)) if rate else 0).reshape(b, t, d)

    def _mlp_branch(self, h, resid, keys, rate):
        """resid + dropout2(dense2(dense1(h))) (layers/transformer.py:60-63,72-76): GELU in dense1's epilogue, dropout + residual in dense2's."""
        b, t, d = resid.shape
        # u stays fp32 ON THE TAPE (dense2 rounds it to bf16 once, as the engine's stored u): its gradient then comes back in fp32 and
        # dense1's backward rounds d(a1) = d(u) * gelu' to bf16 ONCE - the engine's gelu'-multiply epilogue; a bf16 u would force a
        # bf16 d(u) and a second rounding
        u = AG.LinearFn.apply(h.reshape(b * t, d), self.dense1.kernel.value, self.dense1.bias.value, "gelu", False)     # fp32 [B*T, ff]
        return AG.LinearResidualFn.apply(u, self.dense2.kernel.value, self.dense2.bias.value, resid.reshape(b * t, d), rate,
                                         (keys.get("mlp") or _next_key(self._site + 1)) if rate else 0).reshape(b, t, d)

    def call(self, inputs, mask=None, training=None, keys=None, **kwargs):
        keys = keys or {}
        if mask is not None:
            # layers/transformer.py:65-68: self-attention with mask=[mask, mask] (a padding mask over the sequence)
            return self._call_masked(inputs, mask, training, keys)
        rate = self.dense_dropout_rate if training else 0.0
        x = inputs
        if x.dtype != torch.float32:
            x = x.to(torch.float32)
        if not self.pre_norm:
            # the reference's default, layers/transformer.py:59-61: x = norm1(x + attn(x)); x = norm2(x + mlp(x)).  Same kernels as the
            # pre-norm block (the ViT's), composed the other way round.
            x1 = self.norm1(self._attention_branch(x, x, training, keys, rate))            # bf16
            y = self._mlp_branch(x1, x1, keys, rate)
            return AG.CastF32Fn.apply(self.norm2(y))
        x_mid = self._attention_branch(self.norm1(x), x, training, keys, rate)
        return self._mlp_branch(self.norm2(x_mid), x_mid, keys, rate)

    def _call_masked(self, x, mask, training, keys):
        """The block with a sequence mask: the same composition from the stand-alone sub-layers (attention through the general kernel)."""
        rate = self.dense_dropout_rate if training else 0.0
        x = x if x.dtype == torch.float32 else x.to(torch.float32)
        attn = lambda h: self.dropout1(self.multi_head_attention([h, h, h], mask=[mask, mask], training=training, key=keys.get("attn")),   # noqa: E731
                                       training=training, key=keys.get("proj"))
        if self.pre_norm:
            x_mid = AG.AddFn.apply(x, attn(self.norm1(x)))
            return self._mlp_branch(self.norm2(x_mid), x_mid, keys, rate)
        x1 = self.norm1(AG.AddFn.apply(x, attn(x)))
        return AG.CastF32Fn.apply(self.norm2(self._mlp_branch(x1, x1, keys, rate)))

    def get_config(self):
        config = {"embed_dim": self.embed_dim, "num_heads": self.num_heads, "ff_dim": self.ff_dim,
                  "dense_kernel_initializer": self.dense_kernel_initializer, "attention_dropout_rate": self.attention_dropout_rate,
                  "dense_dropout_rate": self.dense_dropout_rate, "norm_epsilon": self.norm_epsilon, "pre_norm": self.pre_norm}
        return dict(list(super(EncoderLayer, self).get_config().items()) + list(config.items()))


@register_keras_serializable(package="Chambers")
class Encoder(Layer):
    def __init__(self, embed_dim, num_heads, ff_dim, num_layers, dense_kernel_initializer="glorot_uniform", attention_dropout_rate=0.1,
                 dense_dropout_rate=0.1, norm_epsilon=1e-6, pre_norm=False, norm_output=False, **kwargs):
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.ff_dim = ff_dim
        self.num_layers = num_layers
        self.dense_kernel_initializer = dense_kernel_initializer
        self.attention_dropout_rate = attention_dropout_rate
        self.dense_dropout_rate = dense_dropout_rate
        self.norm_epsilon = norm_epsilon
        self.pre_norm = pre_norm
        self.norm_output = norm_output
        self.norm_layer = LayerNormalization(epsilon=norm_epsilon) if norm_output else None
        self.layers = []
        self.supports_masking = True
        super(Encoder, self).__init__(**kwargs)

    def _sublayers(self):
        return list(self.layers) + ([self.norm_layer] if self.norm_layer is not None else [])

    def build(self, input_shape):
        self.layers = [EncoderLayer(embed_dim=self.embed_dim, num_heads=self.num_heads, ff_dim=self.ff_dim,
                                    dense_kernel_initializer=self.dense_kernel_initializer,
                                    attention_dropout_rate=self.attention_dropout_rate, dense_dropout_rate=self.dense_dropout_rate,
                                    norm_epsilon=self.norm_epsilon, pre_norm=self.pre_norm) for _ in range(self.num_layers)]
        for layer in self.layers:
            layer.build(tuple(input_shape))
            layer.built = True
        if self.norm_layer is not None:
            self.norm_layer.build(tuple(input_shape))
            self.norm_layer.built = True

    def call(self, inputs, mask=None, training=None, **kwargs):
        x = inputs
        for layer in self.layers:
            x = layer(x, mask=mask, training=training)
        if self.norm_output:
            x = self.norm_layer(x)
        return x

    def get_config(self):
        config = {"embed_dim": self.embed_dim, "num_heads": self.num_heads, "ff_dim": self.ff_dim, "num_layers": self.num_layers,
                  "dense_kernel_initializer": self.dense_kernel_initializer, "attention_dropout_rate": self.attention_dropout_rate,
                  "dense_dropout_rate": self.dense_dropout_rate, "norm_epsilon": self.norm_epsilon, "pre_norm": self.pre_norm,
                  "norm_output": self.norm_output}
        return dict(list(super(Encoder, self).get_config().items()) + list(config.items()))
