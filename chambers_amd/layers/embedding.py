"""chambers.layers.embedding on MI355X: ConcatEmbedding (class token) and LearnedEmbedding1D (positional table),
reference chambers/layers/embedding.py:156-182,218-261.  In the whole-model engine these two are fused into the
patch-embedding GEMM epilogue; the standalone layers below are the API-surface form (pure data movement)."""
import ctypes

import torch

from .._keras_like import Layer, register_keras_serializable
from .. import _lib
from .. import initializers
from .. import kernels as K


@register_keras_serializable(package="Chambers")
class LearnedEmbedding1D(Layer):
    def __init__(self, initializer=None, dtype=None, add_to_input=True, name="learned_embedding", **kwargs):
        self.initializer = initializer
        self.add_to_input = add_to_input
        self.supports_masking = True
        super(LearnedEmbedding1D, self).__init__(dtype=dtype, name=name, **kwargs)

    def build(self, input_shape):
        self.embedding = self.add_weight("embeddings", [input_shape[1], input_shape[-1]], self.initializer)

    def call(self, inputs, **kwargs):
        if self.add_to_input:
            return inputs + self.embedding.value.to(inputs.dtype)
        return self.embedding.value

    def get_config(self):
        config = {"initializer": initializers.serialize(self.initializer), "add_to_input": self.add_to_input}
        return dict(list(super(LearnedEmbedding1D, self).get_config().items()) + list(config.items()))


class LearnedEmbedding0D(LearnedEmbedding1D):
    def build(self, input_shape):
        self.embedding = self.add_weight("embeddings", [1, input_shape[-1]], self.initializer)


@register_keras_serializable(package="Chambers")
class ConcatEmbedding(Layer):
    def __init__(self, n_embeddings, embedding_dim, axis=-1, side="left", initializer=None, dtype=None, name="concat_embedding", **kwargs):
        assert side == "left" or side == "right", "Argument `side` must be either 'left' or 'right'."
        self.n_embeddings = n_embeddings
        self.embedding_dim = embedding_dim
        self.axis = axis
        self.side = side
        self.initializer = initializer
        super(ConcatEmbedding, self).__init__(dtype=dtype, name=name, **kwargs)

    def build(self, input_shape):
        self.embedding = self.add_weight("embeddings", [self.n_embeddings, self.embedding_dim], self.initializer)

    def call(self, inputs, **kwargs):
        batch_size = inputs.shape[0]
        if self.axis in (1, -2) and inputs.dim() == 3 and inputs.is_cuda and inputs.dtype == self.embedding.value.dtype:
            # tf.concat along the token axis as strided row copies (the embedding rows are read with stride 0 over the batch)
            b, n, d = inputs.shape
            es = inputs.element_size()
            out = torch.empty((b, n + self.n_embeddings, d), dtype=inputs.dtype, device=inputs.device)
            emb, x = self.embedding.value.contiguous(), inputs.contiguous()
            e_at, x_at = (0, self.n_embeddings) if self.side == "left" else (n, 0)
            row = (n + self.n_embeddings) * d * es
            _lib.call("chb_copy_rows", _lib.ptr(emb), 0, ctypes.c_void_p(out.data_ptr() + e_at * d * es), row, b, self.n_embeddings * d * es, K._s())
            _lib.call("chb_copy_rows", _lib.ptr(x), n * d * es, ctypes.c_void_p(out.data_ptr() + x_at * d * es), row, b, n * d * es, K._s())
            return out
        embedding = self.embedding.value.to(inputs.dtype).unsqueeze(0).expand(batch_size, self.n_embeddings, self.embedding_dim)
        x = [embedding, inputs] if self.side == "left" else [inputs, embedding]
        return torch.cat(x, dim=self.axis)

    def compute_output_shape(self, input_shape):
        shape = list(input_shape)
        shape[self.axis] = shape[self.axis] + self.n_embeddings
        return tuple(shape)

    def get_config(self):
        config = {"n_embeddings": self.n_embeddings, "embedding_dim": self.embedding_dim, "axis": self.axis, "side": self.side,
                  "initializer": initializers.serialize(self.initializer)}
        return dict(list(super(ConcatEmbedding, self).get_config().items()) + list(config.items()))
