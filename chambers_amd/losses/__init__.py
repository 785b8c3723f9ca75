"""chambers.losses on MI355X: the metric-learning loss the ViT backbones are trained with."""
from .metric_learning import MultiSimilarityLoss  # noqa: F401
