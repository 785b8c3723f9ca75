"""chambers.layers.embedding on MI355X: ConcatEmbedding (class token) and LearnedEmbedding1D (positional table),
reference chambers/layers/embedding.py:156-182,218-261.  In the whole-model engine these two are fused into the
patch-embedding GEMM epilogue; the standalone layers below are the API-surface form (pure data movement)."""
import torch

from .._keras_like import Layer, register_keras_serializable
from .. import initializers
from . import autograd as AG


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
            return AG.AddTableFn.apply(inputs, self.embedding.value)        # fp32 [B, N, D]; d(embeddings) = sum over the batch
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
        if not (self.axis in (1, -2) and inputs.dim() == 3):
            raise ValueError("ConcatEmbedding on MI355X concatenates along the token axis of a [batch, tokens, dim] tensor (axis=1), the "
                             "form the ViT builders use (vision_transformer.py:249-256); got axis=%r on a %d-D input" % (self.axis, inputs.dim()))
        return AG.ConcatTokensFn.apply(inputs, self.embedding.value, self.side == "left")

    def compute_output_shape(self, input_shape):
        shape = list(input_shape)
        shape[self.axis] = shape[self.axis] + self.n_embeddings
        return tuple(shape)

    def get_config(self):
        config = {"n_embeddings": self.n_embeddings, "embedding_dim": self.embedding_dim, "axis": self.axis, "side": self.side,
                  "initializer": initializers.serialize(self.initializer)}
        return dict(list(super(ConcatEmbedding, self).get_config().items()) + list(config.items()))
