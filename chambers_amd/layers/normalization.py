"""chambers.layers.normalization.L2Normalization on MI355X (reference: chambers/layers/normalization.py:5-24):
tf.nn.l2_normalize over the last axis of a [n, d] float tensor, with the backward the metric-learning step needs."""
import torch

from .. import _lib
from .._keras_like import Layer, register_keras_serializable


@register_keras_serializable(package="Chambers")
class L2Normalization(Layer):
    def __init__(self, axis, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis

    def call(self, inputs, **kwargs):
        if inputs.dim() != 2 or self.axis not in (-1, 1):
            raise NotImplementedError("L2Normalization is built for [n, d] embeddings over the last axis")
        _lib.require_gpu(inputs)
        x = inputs.to(torch.float32).contiguous()
        y = torch.empty_like(x)
        inv = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.call("chb_l2_normalize_fwd", _lib.ptr(x), _lib.ptr(y), _lib.ptr(inv), x.shape[0], x.shape[1], torch.cuda.current_stream().cuda_stream)
        self._saved = (y, inv)
        return y

    def backward(self, dy):
        """d(loss)/d(inputs) of the last call."""
        y, inv = self._saved
        dy = dy.to(torch.float32).contiguous()
        dx = torch.empty_like(dy)
        _lib.call("chb_l2_normalize_bwd", _lib.ptr(dy), _lib.ptr(y), _lib.ptr(inv), _lib.ptr(dx), y.shape[0], y.shape[1],
                  torch.cuda.current_stream().cuda_stream)
        return dx

    def get_config(self):
        return dict(super().get_config(), axis=self.axis)
