"""CUTIE base class: loss + inference heads (mirror of /root/reference/model_cutie.py).

``build_loss`` returns the same 5-tuple as the reference (model_cutie.py:140) as symbolic
fetches; the arithmetic (sparse softmax CE, hard negative mining, L2) runs in
cutie_amd.runtime.Engine through cutie_softmax_ce_fwd / cutie_hnm_weights / cutie_l2_add.
"""
from .model_framework import Model, Node


class Fetch(Node):
    """A scalar the train step produces (model_loss, regularization_loss, total_loss)."""

    def __init__(self, name):
        Node.__init__(self, name, "fetch")


def _hp(params, name, default):
    return getattr(params, name) if hasattr(params, name) else default


class CUTIE(Model):
    def __init__(self, num_vocabs, num_classes, params, trainable=True):
        self.name = "CUTIE_benchmark"
        self._init_common(num_vocabs, num_classes, params, trainable, grid_name="data")
        self.data = self.data_grid
        self.layers["data"] = self.data
        self.setup()

    def _init_common(self, num_vocabs, num_classes, params, trainable, grid_name="data_grid"):
        self._init_graph()
        self.data_grid = self.placeholder(grid_name, 1, "grid", "int32")          # [B,H,W,1]
        self.gt_classes = self.placeholder("gt_classes", None, "grid", "int32")   # [B,H,W]
        self.data_image = self.placeholder("data_image", 3, "image", "float32")   # CUTIE2 only
        self.ps_1d_indices = self.placeholder("ps_1d_indices", None, None, "int32")
        self.ghm_weights = self.placeholder("ghm_weights", num_classes, "grid", "float32")
        self.use_ghm = bool(_hp(params, "use_ghm", 0) == 1)
        self.activation = "sigmoid" if _hp(params, "use_ghm", 0) else "relu"
        self.dropout = _hp(params, "data_augmentation_dropout", 1)
        self.layers = dict({grid_name: self.data_grid, "gt_classes": self.gt_classes,
                            "ghm_weights": self.ghm_weights})
        self.num_vocabs = num_vocabs
        self.num_classes = num_classes
        self.trainable = trainable
        self.embedding_size = params.embedding_size
        self.weight_decay = _hp(params, "weight_decay", 0.0)
        self.hard_negative_ratio = _hp(params, "hard_negative_ratio", 0.0)
        self.batch_size = _hp(params, "batch_size", 0)
        self.loss_logits_name = "cls_logits"
        self.loss_spec = {"labels": "gt_classes", "logits": "cls_logits",
                          "hard_negative_ratio": self.hard_negative_ratio, "weight_decay": self.weight_decay}
        # use_ghm = 1 (main_train_json.py:65-67,86-119): the logits layer ends in a sigmoid and the loss is
        # taken of cls_logits * ghm_weights; the weights are computed on the device from the logits
        self.loss_spec.update({"use_ghm": self.use_ghm, "ghm_bins": int(_hp(params, "ghm_bins", 30)),
                               "ghm_momentum": float(_hp(params, "ghm_momentum", 0)), "ghm_eps": float(_hp(params, "eps", 1e-6))})

    def setup(self):
        raise NotImplementedError("the max-pool / up-conv CUTIE_benchmark network (model_cutie.py:32-66) is "
                                  "outside the hot-path scope; use model_cutie_aspp.CUTIERes")

    def inference(self):
        return self.get_output("softmax")

    def build_loss(self):
        """(model_loss, regularization_loss, total_loss, logits, softmax) - model_cutie.py:107-140."""
        self.get_output("gt_classes")
        logits = self.get_output("cls_logits")
        softmax_logits = self.get_output("softmax")
        self.loss_spec.update({"labels": "gt_classes", "logits": "cls_logits",
                               "hard_negative_ratio": self.hard_negative_ratio, "weight_decay": self.weight_decay})
        model_loss = Fetch("model_loss")
        regularization_loss = Fetch("regularization_loss")
        total_loss = Fetch("total_loss")
        return model_loss, regularization_loss, total_loss, logits, softmax_logits
