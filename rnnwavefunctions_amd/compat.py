"""Stand-ins for the handful of TensorFlow-1 objects the reference's call sites touch.

The reference builds graph tensors and evaluates them with ``sess.run``:
    samples_ = wf.sample(numsamples, 2); samples = sess.run(samples_)          (TrainingRNN_1DTFIM.py:189,201)
    ph = tf.placeholder(tf.int32, (None, N)); t = wf.log_probability(ph, 2)
    sess.run(t, feed_dict={ph: chunk})                                         (:192-193, :65)
Here ``wf.sample`` / ``wf.log_probability`` return light *op handles* and ``Session.run`` launches
the HIP kernels, so unmodified estimator call sites keep working.  Ops also evaluate eagerly
(``op.eval()``, ``numpy.asarray(op)``) for code that does not want a session.
"""
import contextlib

import numpy as np

int32, int64, float32, float64, complex64 = np.int32, np.int64, np.float32, np.float64, np.complex64


class Graph:
    """tf.Graph stand-in: only ``as_default()`` is used by the reference (RNNwavefunction.py:28,49)."""

    @contextlib.contextmanager
    def as_default(self):
        yield self


class Placeholder:
    """tf.placeholder stand-in (TrainingRNN_1DTFIM.py:192)."""

    def __init__(self, dtype=np.int32, shape=None, name=None):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else None
        self.name = name

    def __repr__(self):
        return "Placeholder(%s, %s)" % (self.dtype, self.shape)


def placeholder(dtype=np.int32, shape=None, name=None):
    return Placeholder(dtype, shape, name)


class Op:
    """Lazy result of a wave-function method; evaluated by Session.run or .eval()."""

    def __init__(self, wf):
        self.wf = wf

    def eval(self, feed_dict=None):
        return self._run(feed_dict or {})

    def __array__(self, dtype=None, copy=None):
        a = self.eval()
        return a.astype(dtype) if dtype is not None else a


class SampleOp(Op):
    def __init__(self, wf, numsamples):
        super().__init__(wf)
        self.numsamples = int(numsamples)

    def _run(self, feed):
        return self.wf._draw(self.numsamples)


class EvalOp(Op):
    """log_probability / log_amplitude of a fed placeholder (or of a constant array)."""

    def __init__(self, wf, source, kind):
        super().__init__(wf)
        self.source = source
        self.kind = kind          # "log_prob" | "log_amp"

    def _run(self, feed):
        if isinstance(self.source, Placeholder):
            if self.source not in feed:
                raise ValueError("feed_dict lacks a value for %r" % (self.source,))
            x = feed[self.source]
        else:
            x = self.source
        return self.wf._evaluate(np.asarray(x), self.kind)


class ConfigProto:
    """tf.ConfigProto stand-in (TrainingRNN_1DTFIM.py:119-120): attributes are accepted and ignored."""

    class _Opts:
        allow_growth = True

    def __init__(self):
        self.gpu_options = ConfigProto._Opts()


class Session:
    """tf.Session stand-in: ``run(fetches, feed_dict)`` evaluates op handles on the GPU."""

    def __init__(self, graph=None, config=None):
        self.graph = graph
        self.config = config

    def run(self, fetches, feed_dict=None):
        feed = feed_dict or {}
        if isinstance(fetches, (list, tuple)):
            return type(fetches)(self.run(f, feed) for f in fetches)
        if isinstance(fetches, Op):
            return fetches._run(feed)
        if fetches is None:
            return None
        raise TypeError("Session.run: cannot evaluate %r (only ops created by rnnwavefunctions_amd wave "
                        "functions run here; there is no TensorFlow graph and no CPU fallback)" % (fetches,))

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class CudnnCompatibleGRUCell:
    """Sentinel for ``tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell`` (TrainingRNN_1DTFIM.py:103): the GRU
    arithmetic itself lives in the HIP kernels (csrc/gru_core.h)."""

    def __init__(self, num_units=None, *args, **kwargs):
        self.num_units = num_units


def is_gru_cell(cell):
    if cell is None or cell is CudnnCompatibleGRUCell:
        return True
    name = cell if isinstance(cell, str) else getattr(cell, "__name__", type(cell).__name__)
    return "GRU" in str(name).upper()
