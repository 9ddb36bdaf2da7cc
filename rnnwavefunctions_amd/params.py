"""Parameter containers for the RNN wave functions: TensorFlow-style names, initialisers, .npz I/O.

The reference owns its parameters as TF variables created by
``MultiRNNCell([CudnnCompatibleGRUCell(units[n]) ...])`` and ``tf.layers.Dense``
(1DTFIM/RNNwavefunction.py:32-33, J1J2/ComplexRNNwavefunction.py:40-43) or by
``MDRNNcell`` (2DTFIM_2DRNN/MDRNNcell.py:21-35).  Here they are a flat
``{tf_variable_name: ndarray}`` dict so that a TF-side dump maps 1:1
(SURVEY.md 8a row a1/a14, 8b "weight interchange").

Initial values: glorot/xavier-uniform kernels, gate bias 1, other biases 0 (MDRNN: all
five tensors xavier, incl. ``b``), drawn from ``numpy.random.RandomState(seed)``.  TF's own
seeded draws are not reproducible outside TF (SURVEY.md 8c "parity unpinned").
"""
from collections import OrderedDict

import numpy as np

GRU_PREFIX = "multi_rnn_cell/cell_%d/cudnn_compatible_gru_cell/"


def _glorot(rng, shape, dtype):
    if len(shape) == 1:
        fan_in = fan_out = shape[0]            # tf.contrib.layers.xavier_initializer on a vector
    else:
        fan_in, fan_out = shape
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(dtype)


def init_gru_params(units, seed=111, scope="RNNwavefunction", dtype=np.float32, inputdim=2,
                    heads=("wf_dense",)):
    """Parameters of a stacked cuDNN-compatible GRU + ``Dense(2)`` heads.

    heads=("wf_dense",) is the positive RNN (1DTFIM/RNNwavefunction.py:33);
    heads=("wf_dense_ampl", "wf_dense_phase") the complex RNN
    (J1J2/ComplexRNNwavefunction.py:42-43).
    """
    rng = np.random.RandomState(seed)
    p = OrderedDict()
    d = inputdim
    for layer, h in enumerate(units):
        pre = scope + "/" + GRU_PREFIX % layer
        p[pre + "gates/kernel"] = _glorot(rng, (d + h, 2 * h), dtype)
        p[pre + "gates/bias"] = np.ones(2 * h, dtype=dtype)
        p[pre + "candidate/input_projection/kernel"] = _glorot(rng, (d, h), dtype)
        p[pre + "candidate/input_projection/bias"] = np.zeros(h, dtype=dtype)
        p[pre + "candidate/hidden_projection/kernel"] = _glorot(rng, (h, h), dtype)
        p[pre + "candidate/hidden_projection/bias"] = np.zeros(h, dtype=dtype)
        d = h
    for head in heads:
        p[scope + "/" + head + "/kernel"] = _glorot(rng, (d, 2), dtype)
        p[scope + "/" + head + "/bias"] = np.zeros(2, dtype=dtype)
    return p


def init_mdrnn_params(num_units, seed=111, scope="RNNwavefunction", dtype=np.float64, inputdim=2,
                      name="rnn_0"):
    """Parameters of the 2D vanilla cell + Dense(2) (2DTFIM_2DRNN/RNNwavefunction.py:32-33)."""
    rng = np.random.RandomState(seed)
    h = num_units
    p = OrderedDict()
    p[scope + "/Wh_" + name] = _glorot(rng, (h, h), dtype)
    p[scope + "/Uh_" + name] = _glorot(rng, (inputdim, h), dtype)
    p[scope + "/Wv_" + name] = _glorot(rng, (h, h), dtype)
    p[scope + "/Uv_" + name] = _glorot(rng, (inputdim, h), dtype)
    p[scope + "/b_" + name] = _glorot(rng, (h,), dtype)
    p[scope + "/wf_dense/kernel"] = _glorot(rng, (h, 2), dtype)
    p[scope + "/wf_dense/bias"] = np.zeros(2, dtype=dtype)
    return p


def scale_kernels(params, factor):
    """'Trained-like' weights for tests/benches: sharpen the conditionals by scaling kernels."""
    out = OrderedDict()
    for k, v in params.items():
        out[k] = (v * factor).astype(v.dtype) if v.ndim == 2 else v.copy()
    return out


def randomize_biases(params, seed, scale=0.3):
    """Make every bias non-trivial so that tests exercise all bias paths."""
    rng = np.random.RandomState(seed)
    out = OrderedDict()
    for k, v in params.items():
        out[k] = (v + scale * rng.standard_normal(v.shape)).astype(v.dtype) if v.ndim == 1 else v.copy()
    return out


def count_params(params):
    """The number the reference prints at start-up (1DTFIM/TrainingRNN_1DTFIM.py:127-136)."""
    return int(sum(v.size for v in params.values()))


def save_npz(path, params):
    np.savez(path, **{k.replace("/", "|"): v for k, v in params.items()})


def load_npz(path):
    with np.load(path, allow_pickle=False) as f:
        return OrderedDict((k.replace("|", "/"), f[k]) for k in f.files)
