"""Alias named by BASELINE.json's north star (`chambers.models.vit`)."""
from .backbones.vision_transformer import *  # noqa: F401,F403
from .backbones.vision_transformer import ViTB16, ViTB32, ViTL16, ViTL32, ViTS16, VisionTransformer, preprocess_input  # noqa: F401
