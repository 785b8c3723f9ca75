"""chambers.activations (reference: chambers/activations.py:5-56).

`gelu` on a tensor is only the standalone form; inside the ViT block the exact-erf GELU is fused
into the fc1 GEMM epilogue (csrc/gemm.hip, CHB_EPI_GELU) and its derivative into the fc2 dgrad
epilogue (CHB_EPI_DGELU), both using the same formula as the `approximate=False` branch (:46-56)."""
import torch


def gelu(features, approximate=False, name=None):
    """Standalone GELU on a torch tensor (host convenience, not on the hot path)."""
    x = features
    if approximate:
        return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * (x + 0.044715 * torch.pow(x, 3))))
    return 0.5 * x * (1.0 + torch.erf(x / 1.4142135623730951))


gelu.__name__ = "gelu"
