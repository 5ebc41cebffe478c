"""CUTIE2-dilate: mirror of /root/reference/model_cutie2_dilate.py:36-93 - 128-channel 3x3
image encoder with rates 1,1,2,4,8,16, positional sampling, 1x1 fuser, CUTIE-ASPP trunk."""
from .model_cutie2 import CUTIE2 as _CUTIE2Base
from .model_cutie_aspp import CUTIERes


class CUTIE2(_CUTIE2Base):
    def __init__(self, num_vocabs, num_classes, params, trainable=True):
        self.name = "CUTIE2_dilate"
        self._init_common2(num_vocabs, num_classes, params, trainable)
        self.setup()

    trunk = CUTIERes.trunk

    def setup(self):
        (self.feed("data_grid")
             .embed(self.num_vocabs, self.embedding_size, name="embedding", dropout=self.dropout))
        (self.feed("data_image")
             .conv(3, 3, 128, 1, 1, name="image_encoder1_1")
             .conv(3, 3, 128, 1, 1, name="image_encoder1_2")
             .dilate_conv(3, 3, 128, 1, 1, 2, name="image_encoder1_5")
             .dilate_conv(3, 3, 128, 1, 1, 4, name="image_encoder1_6")
             .dilate_conv(3, 3, 128, 1, 1, 8, name="image_encoder1_7")
             .dilate_conv(3, 3, 128, 1, 1, 16, name="image_encoder1_8"))
        (self.feed("image_encoder1_8", "ps_1d_indices", "data_grid")
             .positional_sampling(128, name="positional_sampling"))
        (self.feed("positional_sampling", "embedding")
             .concat(3, name="concat")
             .conv(1, 1, 256, 1, 1, name="feature_fuser"))
        self.trunk("feature_fuser")
