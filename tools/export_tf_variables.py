#!/usr/bin/env python
"""Run this where TensorFlow 1.x is installed (it is not in this repository's environment):

    python export_tf_variables.py <checkpoint prefix, e.g. graph/CUTIE_atrousSPP_d20000c9(r80c80)_iter_40000.ckpt> out.npz

Writes every model variable of a vsymbol/CUTIE checkpoint (main_train_json.py:121-128) into an
.npz keyed by the TensorFlow variable name - '<layer>/weights' (HWIO kernels, the [V, E] embedding
table) and '<layer>/biases' - which cutie_amd.checkpoint.load_npz / Engine.load_params read.
Adam slots, beta powers and global_step are skipped.
"""
import sys

import numpy as np


def main(prefix, out):
    import tensorflow as tf
    reader = tf.train.load_checkpoint(prefix)
    arrays = {}
    for name in sorted(reader.get_variable_to_shape_map()):
        if name.endswith("/weights") or name.endswith("/biases"):
            arrays[name] = reader.get_tensor(name).astype(np.float32)
    np.savez(out, **arrays)
    print("wrote %d variables to %s" % (len(arrays), out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
