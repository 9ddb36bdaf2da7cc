#!/usr/bin/env python3
"""Weight interchange with a TensorFlow-1 run of the reference.

Run THIS script where TensorFlow 1.13 and the reference live (it is not needed, and cannot run, on the MI355X box:
TensorFlow is not part of this build).  The native side reads and writes `.npz` files keyed by the TF variable
names (`RNNwavefunction/multi_rnn_cell/cell_0/cudnn_compatible_gru_cell/gates/kernel`, ...,
`RNNwavefunction/wf_dense/bias`; MDRNN: `RNNwavefunction/Wh_rnn_0`, ...), which is exactly what
`tf.train.Saver` stores in the reference's checkpoints (1DTFIM/TrainingRNN_1DTFIM.py:166,219).

    ckpt -> npz :  tools/tf1_weights_npz.py dump  <checkpoint prefix> weights.npz
    npz  -> ckpt:  tools/tf1_weights_npz.py load  weights.npz <new checkpoint prefix>

`dump` needs no graph: it walks the checkpoint's variable map.  Adam slot variables (`.../Adam`, `.../Adam_1`,
`beta1_power`, `beta2_power`) are skipped - the native drivers keep their own optimizer state.
`load` builds one tf.Variable per array and saves them with the same names, so the reference's
`saver.restore(sess, path + '/' + filename)` (:179-181) picks them up (its optimizer slots start from zero).

On the native side:   wf.restore("weights.npz")   /   rnnwavefunctions_amd.params.save_npz("weights.npz", params)
"""
import sys

import numpy as np


def _is_optimizer_slot(name):
    leaf = name.rsplit("/", 1)[-1]
    return leaf in ("Adam", "Adam_1") or name in ("beta1_power", "beta2_power") or leaf.startswith("beta")


def dump(ckpt_prefix, npz_path):
    import tensorflow as tf
    reader = tf.train.load_checkpoint(ckpt_prefix)
    out = {}
    for name in sorted(reader.get_variable_to_shape_map()):
        if _is_optimizer_slot(name):
            continue
        out[name] = reader.get_tensor(name)
    np.savez(npz_path, **out)
    for k, v in out.items():
        print("%-90s %s %s" % (k, v.dtype, v.shape))
    print("wrote %d arrays (%d parameters) to %s" % (len(out), sum(v.size for v in out.values()), npz_path))


def load(npz_path, ckpt_prefix):
    import tensorflow as tf
    tf1 = getattr(tf, "compat", tf).v1 if hasattr(tf, "compat") and hasattr(tf.compat, "v1") else tf
    data = np.load(npz_path)                      # arrays only; allow_pickle stays False
    graph = tf1.Graph()
    with graph.as_default():
        variables = [tf1.Variable(data[k], name=k) for k in data.files]
        saver = tf1.train.Saver({k: v for k, v in zip(data.files, variables)})
        with tf1.Session(graph=graph) as sess:
            sess.run(tf1.global_variables_initializer())
            saver.save(sess, ckpt_prefix)
    print("wrote %d variables to %s" % (len(data.files), ckpt_prefix))


if __name__ == "__main__":
    if len(sys.argv) != 4 or sys.argv[1] not in ("dump", "load"):
        sys.exit(__doc__)
    (dump if sys.argv[1] == "dump" else load)(sys.argv[2], sys.argv[3])
