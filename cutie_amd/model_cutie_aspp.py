"""CUTIE-ASPP ("CUTIE_atrousSPP"): mirror of /root/reference/model_cutie_aspp.py - same class
name, ctor signature, attributes and layer names; the graph is the reference's :35-74."""
from .model_cutie import CUTIE


class CUTIERes(CUTIE):
    def __init__(self, num_vocabs, num_classes, params, trainable=True):
        self.name = "CUTIE_atrousSPP"
        self._init_common(num_vocabs, num_classes, params, trainable)
        self.setup()

    def trunk(self, first_input):
        """encoder -> ASPP -> decoder -> logits (model_cutie_aspp.py:41-74), shared with CUTIE2."""
        (self.feed(first_input)
             .conv(3, 5, 256, 1, 1, name="encoder1_1")
             .conv(3, 5, 256, 1, 1, name="encoder1_2")
             .conv(3, 5, 256, 1, 1, name="encoder1_3")
             .conv(3, 5, 256, 1, 1, name="encoder1_4")
             .dilate_conv(3, 5, 256, 1, 1, 2, name="encoder1_5")
             .dilate_conv(3, 5, 256, 1, 1, 4, name="encoder1_6")
             .dilate_conv(3, 5, 256, 1, 1, 8, name="encoder1_7")
             .dilate_conv(3, 5, 256, 1, 1, 16, name="encoder1_8"))
        for branch, rate in (("aspp_1", 4), ("aspp_2", 8), ("aspp_3", 16)):
            self.feed("encoder1_8").dilate_conv(3, 5, 256, 1, 1, rate, name=branch)
        self.feed("encoder1_8").global_pool(name="aspp_4")
        (self.feed("aspp_1", "aspp_2", "aspp_3", "aspp_4")
             .concat(3, name="aspp_concat")
             .conv(1, 1, 256, 1, 1, name="aspp_1x1"))
        (self.feed("encoder1_1", "aspp_1x1")
             .concat(3, name="concat1")
             .conv(3, 5, 64, 1, 1, name="decoder1_1"))
        (self.feed("decoder1_1")
             .conv(1, 1, self.num_classes, 1, 1, activation=self.activation, name="cls_logits")
             .softmax(name="softmax"))

    def setup(self):
        (self.feed("data_grid")
             .embed(self.num_vocabs, self.embedding_size, name="embedding", dropout=self.dropout))
        self.trunk("embedding")
