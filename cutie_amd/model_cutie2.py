"""CUTIE2 base (mirror of /root/reference/model_cutie2.py): same loss head as CUTIE
(model_cutie2.py:114-147 duplicates model_cutie.py:107-140)."""
from .model_cutie import CUTIE


class CUTIE2(CUTIE):
    def __init__(self, num_vocabs, num_classes, params, trainable=True):
        self.name = "CUTTIE_benchmark"
        self._init_common2(num_vocabs, num_classes, params, trainable)
        self.setup()

    def _init_common2(self, num_vocabs, num_classes, params, trainable):
        self._init_common(num_vocabs, num_classes, params, trainable)
        self.layers["data_image"] = self.data_image
        self.layers["ps_1d_indices"] = self.ps_1d_indices
