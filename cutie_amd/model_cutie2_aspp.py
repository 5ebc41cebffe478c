"""CUTIE2-ASPP (CUTIE-B): mirror of /root/reference/model_cutie2_aspp.py:36-107 - 3x3 image
branch with its own ASPP, positional sampling of the image feature map at the token
positions, 1x1 fuser, then the CUTIE-ASPP trunk."""
from .model_cutie2 import CUTIE2 as _CUTIE2Base
from .model_cutie_aspp import CUTIERes


class CUTIE2(_CUTIE2Base):
    def __init__(self, num_vocabs, num_classes, params, trainable=True):
        self.name = "CUTIE2_dilate"      # sic: the reference names both CUTIE2 variants this way
        self._init_common2(num_vocabs, num_classes, params, trainable)
        self.setup()

    trunk = CUTIERes.trunk

    def setup(self):
        (self.feed("data_grid")
             .embed(self.num_vocabs, self.embedding_size, name="embedding", dropout=self.dropout))
        (self.feed("data_image")
             .conv(3, 3, 32, 1, 1, name="image_encoder1_1")
             .conv(3, 3, 32, 1, 1, name="image_encoder1_2")
             .dilate_conv(3, 3, 32, 1, 1, 2, name="image_encoder1_5")
             .dilate_conv(3, 3, 32, 1, 1, 2, name="image_encoder1_6")
             .dilate_conv(3, 3, 32, 1, 1, 2, name="image_encoder1_7")
             .dilate_conv(3, 3, 32, 1, 1, 2, name="image_encoder1_8"))
        for branch, rate in (("image_aspp_1", 4), ("image_aspp_2", 8), ("image_aspp_3", 16)):
            self.feed("image_encoder1_8").dilate_conv(3, 3, 32, 1, 1, rate, name=branch)
        self.feed("image_encoder1_8").global_pool(name="image_aspp_4")
        (self.feed("image_aspp_1", "image_aspp_2", "image_aspp_3", "image_aspp_4")
             .concat(3, name="image_aspp_concat")
             .conv(1, 1, 32, 1, 1, name="image_aspp_1x1"))
        (self.feed("image_encoder1_1", "image_aspp_1x1")
             .concat(3, name="image_concat1")
             .conv(3, 3, 32, 1, 1, name="image_featuremap"))
        (self.feed("image_featuremap", "ps_1d_indices", "data_grid")
             .positional_sampling(32, name="positional_sampling"))
        (self.feed("positional_sampling", "embedding")
             .concat(3, name="concat")
             .conv(1, 1, 256, 1, 1, name="feature_fuser"))
        self.trunk("feature_fuser")
