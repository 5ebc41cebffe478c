"""Weight interchange with the reference's TensorFlow checkpoints.

The reference names its variables ``<layer>/weights`` (conv kernels HWIO, model_framework.py:416;
the embedding table [V, E], :47) and ``<layer>/biases`` (:417) and saves them with tf.train.Saver
(main_train_json.py:121-128).  TensorFlow is not available here, so the interchange file is a plain
``.npz`` keyed by exactly those names; ``tools/export_tf_variables.py`` is the few lines a user runs
where TensorFlow exists to turn a checkpoint into such a file (optimizer slots are dropped).
Loaders here never unpickle: ``numpy.load(allow_pickle=False)``.
"""
from collections import OrderedDict

import numpy as np


def expected_variables(network):
    """name -> shape for a cutie_amd model object, in the reference's creation order"""
    return OrderedDict((s.name, tuple(int(d) for d in s.shape)) for s in network.param_specs)


def save_npz(path, named):
    """named: mapping variable name -> array (e.g. Engine.state_dict())."""
    arrays = OrderedDict((k, np.ascontiguousarray(np.asarray(v, dtype=np.float32))) for k, v in named.items())
    np.savez(path, **arrays)


def load_npz(path, network=None, strict=True):
    """Returns an OrderedDict name -> float32 array.  With `network`, checks names and shapes:
    a missing or mis-shaped variable raises; unknown names (optimizer slots, global_step) raise in
    strict mode and are dropped otherwise."""
    out = OrderedDict()
    with np.load(path, allow_pickle=False) as z:
        for k in z.files:
            out[k] = np.asarray(z[k], dtype=np.float32)
    if network is None:
        return out
    want = expected_variables(network)
    extra = [k for k in out if k not in want]
    if extra and strict:
        raise KeyError("%s holds variables the model does not have: %s" % (path, ", ".join(sorted(extra)[:8])))
    missing = [k for k in want if k not in out]
    if missing:
        raise KeyError("%s lacks variables of the model: %s" % (path, ", ".join(missing[:8])))
    for k, shape in want.items():
        if tuple(out[k].shape) != shape:
            raise ValueError("%s: %s has shape %s, the model needs %s" % (path, k, tuple(out[k].shape), shape))
    return OrderedDict((k, out[k]) for k in want)
