"""chambers.activations (reference: chambers/activations.py:5-56).

`gelu` on a tensor is only the standalone form; inside the ViT block the exact-erf GELU is fused
into the fc1 GEMM epilogue (csrc/gemm.hip, CHB_EPI_GELU) and its derivative into the fc2 dgrad
epilogue (CHB_EPI_DGELU), both using the same formula as the `approximate=False` branch (:46-56)."""
def gelu(features, approximate=False, name=None):
    """Stand-alone GELU on a device tensor (fp32 or bf16): chb_gelu_f32 forward, its saved derivative times dy backward
    (layers/autograd.GeluFn) - differentiable, no torch arithmetic."""
    from .layers.autograd import GeluFn
    return GeluFn.apply(features, bool(approximate))


gelu.__name__ = "gelu"
