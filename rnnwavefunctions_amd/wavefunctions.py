"""Host-side wave-function classes shared by the reference-named modules of the sub-packages."""
from collections import OrderedDict

import numpy as np

from . import _lib
from . import params as P
from . import tf_checkpoint as T
from .compat import EvalOp, Graph, Placeholder, SampleOp, is_gru_cell


class _NativeWF:
    """Common part: owns one native handle (one GPU), the TF-named parameter dict and the sampler
    stream position (seed, step)."""

    _model = None

    def _setup(self, nx, ny, units, scope, seed, device, params):
        self.graph = Graph()
        self.scope = scope
        self.seed = int(seed)
        self.units = list(units)
        self._step = 0                    # one Philox sub-stream per drawn batch (sess.run(samples_))
        self._sample_offset = 0           # first GLOBAL sample index of this device's shard
        self._native = _lib.NativeWavefunction(self._model, nx, ny, tuple(units), device=device)
        self.graph.wavefunctions.append(self)
        self.set_params(params)

    # -- parameters (tf.train.Saver stand-in: flat {tf_variable_name: array}) ----------------------
    def set_params(self, params):
        self.params = OrderedDict((k, np.array(v)) for k, v in params.items())
        self._native.set_params(self.params, scope=self.scope)

    def get_params(self):
        return OrderedDict((k, v.copy()) for k, v in self.params.items())

    def save(self, path, extra=None):
        """tf.train.Saver.save stand-in (1DTFIM/TrainingRNN_1DTFIM.py:219): `path` ending in .npz -> NumPy archive
        keyed by the TF variable names; anything else is a TF checkpoint prefix (V2 tensor bundle: <path>.index +
        <path>.data-00000-of-00001, tf_checkpoint.py).  `extra`: further variables of the Saver (optimizer slots, step)."""
        if str(path).endswith(".npz"):
            P.save_npz(path, self.params)
        else:
            T.write_checkpoint(str(path), dict(self.params, **(extra or {})))

    def restore(self, path):
        """tf.train.Saver.restore stand-in (:172-183): loads this model's variables from a .npz or a TF checkpoint
        prefix; optimizer slots / step counters in the file are returned (tf_checkpoint.split_saver_variables), not used."""
        if str(path).endswith(".npz"):
            self.set_params(P.load_npz(path))
            return None
        model, opt = T.split_saver_variables(T.read_checkpoint(str(path)))
        self.set_params(match_checkpoint_names(self.params, model, self.scope))
        return opt

    def num_params(self):
        return self._native.num_params()

    def _views(self, *prefixes):
        """Parameter views by TF-name prefix below the scope: what the reference keeps as layer objects
        (self.rnn = MultiRNNCell(...), self.dense = Dense(...), RNNwavefunction.py:32-33)."""
        pre = tuple(self.scope + "/" + p for p in prefixes)
        return ParameterView(self, [k for k in self.params if k.startswith(pre)])

    def set_shard(self, sample_offset):
        """Multi-GPU: this handle draws global samples [sample_offset, sample_offset + numsamples)."""
        self._sample_offset = int(sample_offset)

    # -- evaluation hooks used by compat.Op ---------------------------------------------------------
    def _draw(self, numsamples):
        s = self._native.sample(numsamples, self.seed, self._step, self._sample_offset)
        self._step += 1
        self._last_drawn = s
        return s.astype(np.int64)          # tf.multinomial returns int64 (RNNwavefunction.py:68)

    def _evaluate(self, x, kind):
        if kind == "log_amp":
            return self._native.log_amp(x)
        return self._native.log_prob(x)

    def _make_eval(self, samples, kind):
        if isinstance(samples, Placeholder):
            return EvalOp(self, samples, kind)
        return self._evaluate(np.asarray(samples), kind)


class ParameterView:
    """Stand-in for a Keras/TF-1 layer object holding some of a wave function's variables: `.variables`,
    `.trainable_variables`, `.weights` (TF names + current values), `.get_weights()` / `.set_weights()`."""

    def __init__(self, wf, keys):
        self._wf, self._keys = wf, list(keys)

    @property
    def variables(self):
        from .compat import VariableRef
        return [VariableRef(self._wf, k) for k in self._keys]

    trainable_variables = weights = trainable_weights = variables

    def get_weights(self):
        return [self._wf.params[k].copy() for k in self._keys]

    def set_weights(self, values):
        if len(values) != len(self._keys):
            raise ValueError("expected %d arrays" % len(self._keys))
        p = self._wf.get_params()
        for k, v in zip(self._keys, values):
            if tuple(np.shape(v)) != p[k].shape:
                raise ValueError("%s: shape %s, expected %s" % (k, np.shape(v), p[k].shape))
            p[k] = np.asarray(v, dtype=p[k].dtype)
        self._wf.set_params(p)

    def count_params(self):
        return int(sum(self._wf.params[k].size for k in self._keys))


def match_checkpoint_names(params, tensors, scope):
    """{our name: checkpoint array} for every parameter: exact name, or the same variable under another scope prefix
    (a checkpoint written with scope='RNNwavefunction' restores into scope='myscope').  Shapes and dtypes must agree."""
    out = OrderedDict()
    by_tail = {}
    for name, a in tensors.items():
        by_tail.setdefault(name.split("/", 1)[-1], []).append((name, a))
    for k, v in params.items():
        if k in tensors:
            a = tensors[k]
        else:
            cands = by_tail.get(k[len(scope) + 1:] if k.startswith(scope + "/") else k, [])
            if len(cands) != 1:
                raise KeyError("checkpoint has %s variable named like %r (have: %s)"
                               % ("no" if not cands else "more than one", k, sorted(tensors)[:6]))
            a = cands[0][1]
        if tuple(a.shape) != tuple(v.shape):
            raise ValueError("%s: checkpoint shape %s, model shape %s" % (k, a.shape, v.shape))
        if a.dtype != v.dtype:
            raise ValueError("%s: checkpoint dtype %s, model dtype %s" % (k, a.dtype, v.dtype))
        out[k] = a
    return out


class GRUWavefunction1D(_NativeWF):
    """1DTFIM/RNNwavefunction.py:7-118 - positive RNN wave function (stacked GRU -> Dense(2) softmax)."""

    _model = _lib.MODEL_GRU1D
    _dtype = np.float32
    _heads = ("wf_dense",)

    def __init__(self, systemsize, cell=None, units=[10], scope="RNNwavefunction", seed=111, device=0):
        if not is_gru_cell(cell):
            raise ValueError("only the cuDNN-compatible GRU cell of the reference's run scripts is implemented "
                             "(got cell=%r)" % (cell,))
        self.N = systemsize
        prm = P.init_gru_params(units, seed=seed, scope=scope, dtype=self._dtype, heads=self._heads)
        self._setup(systemsize, 1, units, scope, seed, device, prm)
        self._layer_views()

    def _layer_views(self):
        self.rnn = self._views("multi_rnn_cell/")                         # RNNwavefunction.py:32
        if len(self._heads) == 1:
            self.dense = self._views("wf_dense/")                         # :33
        else:
            self.dense_ampl = self._views("wf_dense_ampl/")               # ComplexRNNwavefunction.py:42-43
            self.dense_phase = self._views("wf_dense_phase/")

    def sample(self, numsamples, inputdim):
        """RNNwavefunction.sample (:35-74): op handle yielding (numsamples, N) int64 spins."""
        self._check_inputdim(inputdim)
        self.inputdim = self.outputdim = inputdim
        self.numsamples = numsamples
        self.samples = SampleOp(self, numsamples)
        return self.samples

    def log_probability(self, samples, inputdim):
        """RNNwavefunction.log_probability (:76-118): f64 (B,) log-probabilities."""
        self._check_inputdim(inputdim)
        self.inputdim = self.outputdim = inputdim
        self.log_probs = self._make_eval(samples, "log_prob")
        return self.log_probs

    @staticmethod
    def _check_inputdim(inputdim):
        if int(inputdim) != 2:
            raise ValueError("spin-1/2 only: inputdim must be 2 (as in every reference run script)")


class GRUWavefunction1DParity(GRUWavefunction1D):
    """1DTFIM/RNNwavefunction_paritysym.py:7-145 - same sampler, log P symmetrised under reflection."""

    _model = _lib.MODEL_GRU1D_PARITY


class ComplexGRUWavefunction1D(GRUWavefunction1D):
    """J1J2/ComplexRNNwavefunction.py:15-169 - complex RNN with the U(1) zero-magnetisation mask."""

    _model = _lib.MODEL_CRNN_U1
    _heads = ("wf_dense_ampl", "wf_dense_phase")

    def __init__(self, systemsize, cell=None, units=[10, 10], scope="RNNwavefunction", seed=111, device=0):
        super().__init__(systemsize, cell=cell, units=units, scope=scope, seed=seed, device=device)

    def log_amplitude(self, samples, inputdim):
        """ComplexRNNwavefunction.log_amplitude (:105-169): complex64 (B,) log-amplitudes."""
        self._check_inputdim(inputdim)
        self.inputdim = self.outputdim = inputdim
        self.log_amplitudes = self._make_eval(samples, "log_amp")
        return self.log_amplitudes


class GRUWavefunction2DRaster(GRUWavefunction1D):
    """2DTFIM_1DRNN/RNNwavefunction.py:8-130 - float64 GRU run over the flattened lattice."""

    _model = _lib.MODEL_GRU1D_F64
    _dtype = np.float64

    def __init__(self, systemsize_x, systemsize_y, cell=None, units=[10], scope="RNNwavefunction", seed=111,
                 activation=None, device=0):
        if not is_gru_cell(cell):
            raise ValueError("only the GRU cell the reference's training script passes is implemented "
                             "(got cell=%r)" % (cell,))
        self.Nx, self.Ny = systemsize_x, systemsize_y
        self.N = systemsize_x * systemsize_y
        prm = P.init_gru_params(units, seed=seed, scope=scope, dtype=np.float64)
        self._setup(systemsize_x, systemsize_y, units, scope, seed, device, prm)
        self._layer_views()


class MDRNNWavefunction2D(_NativeWF):
    """2DTFIM_2DRNN/RNNwavefunction.py:5-200 - zig-zag 2D RNN wave function, float64."""

    _model = _lib.MODEL_MDRNN2D

    def __init__(self, systemsize_x, systemsize_y, cell=None, units=[10], scope="RNNwavefunction", seed=111, device=0):
        name = getattr(cell, "__name__", "") if cell is not None else "MDRNNcell"
        if "MDRNN" not in name:
            raise ValueError("the 2D wave function runs the MDRNNcell only (got cell=%r)" % (cell,))
        self.Nx, self.Ny = systemsize_x, systemsize_y
        self.rnn = cell(num_units=units[0], num_in=2, name="rnn_0", dtype=np.float64) if cell is not None else None
        prm = P.init_mdrnn_params(units[0], seed=seed, scope=scope)
        self._setup(systemsize_x, systemsize_y, units[:1], scope, seed, device, prm)
        self.cell = self._views("Wh_", "Uh_", "Wv_", "Uv_", "b_")        # the MDRNNcell's variables (MDRNNcell.py:21-35)
        self.dense = self._views("wf_dense/")                             # 2DTFIM_2DRNN/RNNwavefunction.py:33

    def sample(self, numsamples, inputdim):
        GRUWavefunction1D._check_inputdim(inputdim)
        self.inputdim = self.outputdim = inputdim
        self.numsamples = numsamples
        self.samples = SampleOp(self, numsamples)
        return self.samples

    def log_probability(self, samples, inputdim):
        GRUWavefunction1D._check_inputdim(inputdim)
        self.inputdim = self.outputdim = inputdim
        self.log_probs = self._make_eval(samples, "log_prob")
        return self.log_probs
