"""chambers.losses namespace (metric-learning pair losses; reference: chambers/losses/metric_learning.py)."""
from .metric_learning import ContrastiveLoss, MultiSimilarityLoss, MultiSimilarityLossMatrix, NTXentLoss  # noqa: F401
