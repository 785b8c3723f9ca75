from .vision_transformer import ViTS16, ViTB16, ViTB32, ViTL16, ViTL32, VisionTransformer  # noqa: F401
