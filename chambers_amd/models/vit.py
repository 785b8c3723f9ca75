"""Alias named by BASELINE.json's north star (`chambers.models.vit`)."""
from .backbones.vision_transformer import *  # noqa: F401,F403
from .backbones.vision_transformer import (DeiTB16, DeiTS16, DistilledVisionTransformer, ViTB16, ViTB32, ViTL16, ViTL32, ViTS16,  # noqa: F401
                                           VisionTransformer, preprocess_input)
