"""chambers.miners.MultiSimilarityMiner (reference: chambers/miners.py:48-60): keeps the positives that are harder than the
hardest negative plus a margin and the negatives harder than the easiest positive minus the margin.  On MI355X the mining is
fused into the loss kernel (chambers_amd/csrc/metric.hip); this class carries its configuration."""


class MultiSimilarityMiner:
    def __init__(self, margin, name="multi_similarity_miner"):
        self.margin = float(margin)
        self.name = name

    def get_config(self):
        return {"name": self.name, "margin": self.margin}

    @classmethod
    def from_config(cls, config):
        return cls(**config)
