"""Layer DSL with the surface of the reference's model_framework.Model
(/root/reference/model_framework.py): ``feed(...)``, chainable ``@layer`` ops that return
``self`` and register their output under ``self.layers[name]``, ``get_output(name)``,
``get_unique_name(prefix)``, same op names, argument order and layer names.

The reference builds a TensorFlow graph; this builds a small symbolic graph of ``Node``s that
cutie_amd.runtime.Engine compiles into HIP kernel launches.  No arithmetic happens here.

Ops on the hot path (reached from model_cutie_aspp / model_cutie2_aspp / model_cutie2_dilate):
embed, conv, dilate_conv, dilate_module, concat, global_pool, softmax, positional_sampling.
The others the reference defines (bert_embed, sepconv, up_sepconv, dense_block, up_conv,
attention, add, max_pool) are outside the hot-path scope and raise
NotImplementedError by name.
"""

__all__ = ["layer", "Model", "Node", "ParamSpec"]


class Node(object):
    """A symbolic tensor: the stand-in for the tf.Tensor the reference stores in self.layers."""
    __slots__ = ("name", "op", "inputs", "attrs", "channels", "domain")

    def __init__(self, name, op, inputs=(), attrs=None, channels=None, domain=None):
        self.name = name
        self.op = op
        self.inputs = list(inputs)
        self.attrs = dict(attrs or {})
        self.channels = channels
        self.domain = domain      # 'grid' | 'image' | None: which H x W the tensor lives on

    def __repr__(self):
        return "<Node %s %s c=%s>" % (self.name, self.op, self.channels)


class ParamSpec(object):
    """A trainable variable: '<scope>/weights' (HWIO, or [V,E] for the embedding), '<scope>/biases'."""
    __slots__ = ("name", "shape", "init", "regularized", "trainable")

    def __init__(self, name, shape, init, regularized, trainable):
        self.name, self.shape, self.init = name, tuple(int(s) for s in shape), init
        self.regularized, self.trainable = regularized, trainable


def layer(op):
    """Decorator mirroring model_framework.py:7-23: auto name, input unpacking, registration."""
    def layer_decorated(self, *args, **kwargs):
        name = kwargs.setdefault("name", self.get_unique_name(op.__name__))
        if len(self.layer_inputs) == 0:
            raise RuntimeError("No input variables found for layers %s" % name)
        elif len(self.layer_inputs) == 1:
            layer_input = self.layer_inputs[0]
        else:
            layer_input = list(self.layer_inputs)
        layer_output = op(self, layer_input, *args, **kwargs)
        layer_output.name = name
        self.layers[name] = layer_output
        self.nodes.append(layer_output)
        self.feed(layer_output)
        return self
    layer_decorated.__name__ = op.__name__
    return layer_decorated


def _out_of_scope(opname):
    def op(self, *a, **k):
        raise NotImplementedError(
            "Model.%s is outside the MI355X hot-path scope (SURVEY.md section 2: not reached from "
            "model_cutie_aspp / model_cutie2_aspp / model_cutie2_dilate)" % opname)
    op.__name__ = opname
    return op


class Model(object):
    def __init__(self, trainable=True):
        self.layers = dict()
        self.trainable = trainable
        self._init_graph()
        self.setup()

    # -- bookkeeping shared by subclasses that (like the reference's) skip Model.__init__
    def _init_graph(self):
        self.layer_inputs = []
        self.nodes = []         # op nodes in creation (= topological) order
        self.param_specs = []   # variables in creation order (tf.trainable_variables() order)

    def placeholder(self, name, channels, domain, dtype):
        n = Node(name, "input", attrs={"dtype": dtype}, channels=channels, domain=domain)
        return n

    def build_loss(self):
        raise NotImplementedError("Must be subclassed.")

    def setup(self):
        raise NotImplementedError("Must be subclassed.")

    # ------------------------------------------------------------------ hot-path ops
    @layer
    def embed(self, layer_input, vocabulary_size, embedding_size, name, dropout=1, trainable=True):
        if dropout != 1:
            # data_augmentation_dropout is a keep-prob that the reference leaves at 1 (identity)
            raise NotImplementedError("embedding dropout keep_prob %r != 1 is not built" % (dropout,))
        self.make_var(name + "/weights", [vocabulary_size, embedding_size], "uniform(-1,1)", None, trainable)
        return Node(name, "embed", [layer_input], {"V": vocabulary_size, "E": embedding_size},
                    channels=embedding_size, domain="grid")

    @layer
    def positional_sampling(self, layer_input, feature_dimension, name="positional_sampling"):
        featuremap, batch_indices, grid = layer_input[0], layer_input[1], layer_input[2]
        return Node(name, "positional_sampling", [featuremap, batch_indices, grid], {},
                    channels=feature_dimension, domain="grid")

    def _conv_node(self, layer_input, k_h, k_w, c_o, s_h, s_w, rate, name, activation, trainable):
        if (s_h, s_w) != (1, 1):
            raise NotImplementedError("strided convolution is outside the hot-path scope (stride %r)" % ((s_h, s_w),))
        if activation not in ("relu", "sigmoid"):
            activation = "relu"    # model_framework.py:173-175: anything but 'sigmoid' is relu
        if isinstance(layer_input, list):
            raise ValueError("conv takes one input, got %d" % len(layer_input))
        c_i = layer_input.channels
        self.convolution(None, activation, layer_input, k_h, k_w, c_i, c_o, "truncated_normal(0,0.01)", "zeros",
                         True, trainable, scope=name)
        return Node(name, "conv", [layer_input],
                    {"kh": k_h, "kw": k_w, "rate": rate, "activation": activation, "cin": c_i},
                    channels=c_o, domain=layer_input.domain)

    @layer
    def conv(self, layer_input, k_h, k_w, c_o, s_h, s_w, name, activation="relu", trainable=True):
        return self._conv_node(layer_input, k_h, k_w, c_o, s_h, s_w, 1, name, activation, trainable)

    @layer
    def dilate_conv(self, layer_input, k_h, k_w, c_o, s_h, s_w, rate, name, activation="relu", trainable=True):
        return self._conv_node(layer_input, k_h, k_w, c_o, s_h, s_w, rate, name, activation, trainable)

    @layer
    def dilate_module(self, layer_input, k_h, k_w, c_o, s_h, s_w, rate, name, activation="relu", trainable=True):
        return self._conv_node(layer_input, k_h, k_w, c_o, s_h, s_w, rate, name, activation, trainable)

    @layer
    def concat(self, layer_input, axis, name):
        if axis != 3:
            raise NotImplementedError("concat along axis %r (only the channel axis 3 is on the hot path)" % (axis,))
        return Node(name, "concat", list(layer_input), {}, channels=sum(n.channels for n in layer_input),
                    domain=layer_input[0].domain)

    @layer
    def global_pool(self, layer_input, name):
        return Node(name, "global_pool", [layer_input], {}, channels=layer_input.channels,
                    domain=layer_input.domain)

    @layer
    def softmax(self, layer_input, name):
        return Node(name, "softmax", [layer_input], {}, channels=layer_input.channels, domain=layer_input.domain)

    # ------------------------------------------------------------------ out of scope
    bert_embed = _out_of_scope("bert_embed")
    sepconv = _out_of_scope("sepconv")
    up_sepconv = _out_of_scope("up_sepconv")
    dense_block = _out_of_scope("dense_block")
    up_conv = _out_of_scope("up_conv")
    attention = _out_of_scope("attention")
    add = _out_of_scope("add")
    max_pool = _out_of_scope("max_pool")

    # ------------------------------------------------------------------ variables
    def convolution(self, convolve, activate, input, k_h, k_w, c_i, c_o, init_weights, init_biases,
                    regularizer, trainable, name="", scope=""):
        """Registers '<scope>/weights<name>' [k_h,k_w,c_i,c_o] and '<scope>/biases<name>' [c_o]
        (model_framework.py:414-417).  The arithmetic - conv, +bias, activation, instance norm -
        is the Engine's conv_fwd + in_apply launch pair."""
        self.make_var(scope + "/weights" + name, [k_h, k_w, c_i, c_o], init_weights, regularizer, trainable)
        self.make_var(scope + "/biases" + name, [c_o], init_biases, None, trainable)

    def l2_regularizer(self, weight_decay=0.0005, scope=None):
        return ("l2", weight_decay)

    def make_var(self, name, shape, initializer=None, regularizer=None, trainable=True):
        spec = ParamSpec(name, shape, initializer, bool(regularizer), trainable)
        self.param_specs.append(spec)
        return spec

    # ------------------------------------------------------------------ graph plumbing
    def feed(self, *args):
        assert len(args) != 0
        self.layer_inputs = []
        for item in args:
            if isinstance(item, str):
                try:
                    item = self.layers[item]
                except KeyError:
                    print(list(self.layers.keys()))
                    raise KeyError("Unknown layer name fed: %s" % item)
            self.layer_inputs.append(item)
        return self

    def get_output(self, layer):
        try:
            return self.layers[layer]
        except KeyError:
            print(list(self.layers.keys()))
            raise KeyError("Unknown layer name fed: %s" % layer)

    def get_unique_name(self, prefix):
        n = sum(1 for t in self.layers if t.startswith(prefix)) + 1
        return "%s_%d" % (prefix, n)
