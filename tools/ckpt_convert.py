#!/usr/bin/env python3
"""TF checkpoint (V2 tensor bundle) <-> .npz, without TensorFlow (rnnwavefunctions_amd/tf_checkpoint.py).

    python tools/ckpt_convert.py list  path/to/model.ckpt            # names, shapes, dtypes (tf.train.list_variables)
    python tools/ckpt_convert.py npz   path/to/model.ckpt out.npz    # every variable; '/' in names stays (load with params.load_npz)
    python tools/ckpt_convert.py ckpt  weights.npz path/to/out.ckpt  # an .npz written by params.save_npz / wf.save(".npz")

Replaces round 1's TF-side script: a checkpoint the reference's `saver.save` wrote converts here directly, and a
checkpoint written here restores in TF with `tf.train.Saver(var_list=...)` over the model variables it holds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnwavefunctions_amd import params as P            # noqa: E402
from rnnwavefunctions_amd import tf_checkpoint as T     # noqa: E402


def main(argv):
    if len(argv) < 3 or argv[1] not in ("list", "npz", "ckpt"):
        print(__doc__)
        return 2
    if argv[1] == "list":
        for name, shape, dtype in T.list_variables(argv[2]):
            print("%-90s %-14s %s" % (name, shape, getattr(dtype, "__name__", dtype)))
    elif argv[1] == "npz":
        P.save_npz(argv[3], T.read_checkpoint(argv[2]))
        print("wrote", argv[3])
    else:
        T.write_checkpoint(argv[3], P.load_npz(argv[2]))
        print("wrote", argv[3] + ".index", "and", argv[3] + ".data-00000-of-00001")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
