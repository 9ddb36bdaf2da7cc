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


_graph_stack = []


class Graph:
    """tf.Graph stand-in: ``as_default()`` (RNNwavefunction.py:28,49) and the list of wave functions built in it, which
    is what ``tf.trainable_variables()`` / ``tf.global_variables_initializer()`` refer to inside the context."""

    def __init__(self):
        self.wavefunctions = []
        self.opt_steps = []           # training ops built in this graph (apply_gradients): what a Saver keeps of the optimizer

    @contextlib.contextmanager
    def as_default(self):
        _graph_stack.append(self)
        try:
            yield self
        finally:
            _graph_stack.pop()


def get_default_graph():
    return _graph_stack[-1] if _graph_stack else None


class Sym:
    """Just enough symbolic arithmetic to write the reference's VMC cost with these objects
    (`tf.reduce_mean(tf.multiply(log_probs_, Eloc)) - tf.reduce_mean(Eloc) * tf.reduce_mean(log_probs_)`,
    TrainingRNN_1DTFIM.py:156; the complex form at TrainingRNN_J1J2.py:197).  Nothing is evaluated symbolically: the tree
    is only RECOGNISED by `AdamOptimizer.compute_gradients`, whose gradient is rnnwf_vmc_gradient."""

    def __mul__(self, other):
        return Expr("mul", self, other)

    __rmul__ = __mul__

    def __sub__(self, other):
        return Expr("sub", self, other)

    def __rsub__(self, other):
        return Expr("sub", other, self)


class Expr(Sym):
    def __init__(self, op, *args):
        self.op, self.args = op, args


def _sig(e):
    """Canonical string of a cost tree: L = log-probability / log-amplitude op, E = fed local energies."""
    if isinstance(e, Expr):
        if e.op == "stop":
            return _sig(e.args[0])
        parts = [_sig(a) for a in e.args]
        if e.op == "mul":
            parts.sort()
        return "%s(%s)" % (e.op, ",".join(parts))
    if isinstance(e, EvalOp):
        return "L"
    if isinstance(e, Placeholder):
        return "E"
    if isinstance(e, (int, float)):
        return repr(float(e)) if float(e) != int(e) else str(int(e))
    raise TypeError("cannot take %r into a cost expression" % (e,))


def _leaves(e, kind, out):
    if isinstance(e, Expr):
        for a in e.args:
            _leaves(a, kind, out)
    elif isinstance(e, kind) and e not in out:
        out.append(e)
    return out


reduce_mean = lambda x, *a, **k: Expr("mean", x)                 # noqa: E731
multiply = lambda a, b, *r, **k: Expr("mul", a, b)               # noqa: E731
stop_gradient = lambda x, *a, **k: Expr("stop", x)               # noqa: E731
conj = lambda x, *a, **k: Expr("conj", x)                        # noqa: E731
real = lambda x, *a, **k: Expr("real", x)                        # noqa: E731

_COST_REAL = "sub(mean(mul(E,L)),mul(mean(E),mean(L)))"
_COST_COMPLEX = "mul(2,real(sub(mean(mul(E,conj(L))),mul(conj(mean(L)),mean(E)))))"


class Placeholder(Sym):
    """tf.placeholder stand-in (TrainingRNN_1DTFIM.py:192)."""

    def __init__(self, dtype=np.int32, shape=None, name=None):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else None
        self.name = name

    def __repr__(self):
        return "Placeholder(%s, %s)" % (self.dtype, self.shape)


def placeholder(dtype=np.int32, shape=None, name=None):
    return Placeholder(dtype, shape, name)


class Op(Sym):
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
        if isinstance(fetches, VariableRef):
            return fetches.value().copy()
        if isinstance(fetches, str):                  # a variable name, as `sess.run(variables_names)` passes (:130)
            for wf in (self.graph.wavefunctions if self.graph is not None else []):
                if fetches.rsplit(":", 1)[0] in wf.params:
                    return wf.params[fetches.rsplit(":", 1)[0]].copy()
            raise TypeError("Session.run: %r is neither an op of this package nor the name of a variable in this session's graph" % fetches)
        if isinstance(fetches, Variable):
            return fetches.value
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


class _Namespace:
    def __init__(self, **kw):
        self.__dict__.update(kw)


# tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell, the cell every run script of the reference passes
# (1DTFIM/TrainingRNN_1DTFIM.py:103, J1J2/TrainingRNN_J1J2.py:153, 2DTFIM_1DRNN/Training1DRNN_2DTFIM.py:104)
contrib = _Namespace(cudnn_rnn=_Namespace(CudnnCompatibleGRUCell=CudnnCompatibleGRUCell),
                     rnn=_Namespace(GRUCell=CudnnCompatibleGRUCell))
AUTO_REUSE = "AUTO_REUSE"


@contextlib.contextmanager
def variable_scope(name_or_scope=None, reuse=None, **kwargs):
    """tf.variable_scope(wf.scope, reuse=tf.AUTO_REUSE) (TrainingRNN_1DTFIM.py:147,185): parameters live in the wave
    function object, there is nothing to scope."""
    yield name_or_scope


def reset_default_graph():
    del _graph_stack[:]


def set_random_seed(seed):
    """tf.set_random_seed (:85).  The native sampler is seeded per wave function (its `seed` argument); kept so that
    the call site runs."""


class Variable:
    """tf.Variable(0, trainable=False): the global step of the reference's learning-rate schedule (:110)."""

    def __init__(self, initial_value=0, trainable=True, name=None, dtype=None):
        self.value = np.asarray(initial_value, dtype=dtype)
        self.trainable = trainable
        self.name = (name or "Variable") + ":0"


class VariableRef:
    """One TF-named parameter of a wave function, as tf.trainable_variables() lists it (:125-136)."""

    def __init__(self, wf, key):
        self.wf, self.key = wf, key
        self.name = key + ":0"

    def value(self):
        return self.wf.params[self.key]


def trainable_variables():
    g = get_default_graph()
    if g is None:
        raise RuntimeError("tf.trainable_variables(): call it inside `with wf.graph.as_default():`")
    return [VariableRef(wf, k) for wf in g.wavefunctions for k in wf.params]


def reshape(tensor, shape):
    return np.reshape(np.asarray(tensor), shape)


class _NoOp(Op):
    def __init__(self):
        Op.__init__(self, None)

    def _run(self, feed):
        return None


def global_variables_initializer():
    """Parameters are initialised when the wave function is built (params.init_*): running this op does nothing."""
    return _NoOp()


def variables_initializer(var_list=None):
    return _NoOp()


class _Logging:
    ERROR, WARN, INFO, DEBUG = 40, 30, 20, 10

    @staticmethod
    def set_verbosity(level):
        pass


logging = _Logging()


class _LearningRate:
    """tf.train.exponential_decay(lr, global_step, decay_steps, decay_rate, staircase) (:112)."""

    def __init__(self, learning_rate, global_step, decay_steps, decay_rate, staircase=False):
        self.lr, self.step, self.decay_steps, self.decay_rate, self.staircase = learning_rate, global_step, decay_steps, decay_rate, staircase

    def value(self, feed):
        lr = feed[self.lr] if isinstance(self.lr, Placeholder) else self.lr
        t = float(np.asarray(self.step.value if isinstance(self.step, Variable) else self.step))
        e = t / self.decay_steps
        return float(lr) * self.decay_rate ** (np.floor(e) if self.staircase else e)


class _CostPlan:
    def __init__(self, wf, samples_ph, eloc_ph, is_complex):
        self.wf, self.samples_ph, self.eloc_ph, self.is_complex = wf, samples_ph, eloc_ph, is_complex


class _Gradient:
    def __init__(self, plan, var):
        self.plan, self.var = plan, var


class _OptStep(Op):
    """One training step of the reference's loop: `sess.run(optstep, feed_dict=...)` (TrainingRNN_1DTFIM.py:221)."""

    def __init__(self, optimizer, plan, global_step):
        Op.__init__(self, plan.wf)
        self.opt, self.plan, self.global_step = optimizer, plan, global_step

    def _run(self, feed):
        from .training import Adam, cost_gradient
        plan, wf = self.plan, self.plan.wf
        for ph in (plan.samples_ph, plan.eloc_ph):
            if ph not in feed:
                raise ValueError("feed_dict lacks a value for %r" % (ph,))
        samples, eloc = np.asarray(feed[plan.samples_ph]), np.asarray(feed[plan.eloc_ph])
        ns = samples.shape[0]
        lr = self.opt.learning_rate
        lr = lr.value(feed) if isinstance(lr, _LearningRate) else float(feed[lr]) if isinstance(lr, Placeholder) else float(lr)
        wf._native.load_batch(samples, eloc)
        mean_e = complex(np.mean(eloc)) if plan.is_complex else float(np.mean(eloc))
        grads = cost_gradient(wf._native, wf.params, wf.scope, mean_e, ns)
        if self.opt._adam is None:
            self.opt._adam = Adam(self.opt.beta1, self.opt.beta2, self.opt.epsilon)
        wf.set_params(self.opt._adam.step(wf.get_params(), grads, lr))
        if isinstance(self.global_step, Variable):
            self.global_step.value = np.asarray(int(self.global_step.value) + 1)
        return None


class _Train:
    exponential_decay = staticmethod(lambda learning_rate, global_step, decay_steps, decay_rate, staircase=False, name=None:
                                     _LearningRate(learning_rate, global_step, decay_steps, decay_rate, staircase))

    class AdamOptimizer:
        """tf.train.AdamOptimizer(learning_rate, beta1, beta2, epsilon) (:114) for the reference's VMC cost.

        `compute_gradients(cost)` recognises the cost  mean(log P * E) - mean(E) mean(log P)  (TrainingRNN_1DTFIM.py:156,
        with or without stop_gradient on E as in the 2D scripts) and its complex form (TrainingRNN_J1J2.py:197); its
        gradient is rnnwf_vmc_gradient on the fed batch (rnnwf_load_batch).  `apply_gradients(...)` returns the op that
        `sess.run(optstep, feed_dict={Eloc: ..., samp: ..., learningrate_placeholder: lr})` runs: load the batch, take the
        gradient on the GPU, Adam update (training.Adam = TF-1 formulas), advance `global_step`.  Any other graph is
        refused: there is no general autodiff here."""

        def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, name="Adam"):
            self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon
            self._adam = None

        def variables(self):
            return []

        def compute_gradients(self, cost, var_list=None):
            sig = _sig(cost) if isinstance(cost, Expr) else None
            if sig not in (_COST_REAL, _COST_COMPLEX):
                raise NotImplementedError(
                    "compute_gradients: only the reference's VMC cost is differentiated here (rnnwf_vmc_gradient); got %s" % sig)
            ops, phs = _leaves(cost, EvalOp, []), _leaves(cost, Placeholder, [])
            if len(ops) != 1 or len(phs) != 1 or not isinstance(ops[0].source, Placeholder):
                raise NotImplementedError("compute_gradients: expected ONE log-probability op of a placeholder and ONE energy placeholder")
            if (sig == _COST_COMPLEX) != (ops[0].kind == "log_amp"):
                raise NotImplementedError("compute_gradients: the complex cost goes with log_amplitude, the real one with log_probability")
            plan = _CostPlan(ops[0].wf, ops[0].source, phs[0], sig == _COST_COMPLEX)
            return [(_Gradient(plan, v), v) for v in (var_list or [VariableRef(plan.wf, k) for k in plan.wf.params])]

        def apply_gradients(self, grads_and_vars, global_step=None, name=None):
            gv = list(grads_and_vars)
            if not gv or not isinstance(gv[0][0], _Gradient):
                raise NotImplementedError("apply_gradients: pass what compute_gradients returned")
            step = _OptStep(self, gv[0][0].plan, global_step)
            g = get_default_graph()
            if g is not None:
                g.opt_steps.append(step)
            return step

        def minimize(self, cost, global_step=None, var_list=None):
            return self.apply_gradients(self.compute_gradients(cost, var_list), global_step)

    class Saver:
        """tf.train.Saver() (:166): save / restore the wave functions of the graph as TF checkpoints (V2 tensor bundle,
        tf_checkpoint.py) - and, like the reference's Saver (built after apply_gradients), the Adam slots, beta powers and
        global step of the training ops built in the same graph, so that the reference's restore branch (:172-183)
        resumes with its optimizer state instead of restarting Adam at t = 0."""

        def __init__(self, var_list=None):
            g = get_default_graph()
            self.wavefunctions = list(g.wavefunctions) if g is not None else []
            self.opt_steps = list(g.opt_steps) if g is not None else []

        def _wfs(self, sess):
            return self.wavefunctions or (sess.graph.wavefunctions if getattr(sess, "graph", None) else [])

        def _step_of(self, wf):
            for st in self.opt_steps:
                if st.plan.wf is wf:
                    return st
            return None

        def save(self, sess, save_path, global_step=None):
            from . import tf_checkpoint as T
            from .training import Adam
            tensors = {}
            for wf in self._wfs(sess):
                tensors.update(wf.params)
                st = self._step_of(wf)
                if st is not None:
                    adam = st.opt._adam or Adam(st.opt.beta1, st.opt.beta2, st.opt.epsilon)
                    tensors.update(adam.state_tensors(wf.params, wf.scope))
                    if isinstance(st.global_step, Variable):
                        tensors["Variable"] = np.asarray(int(st.global_step.value), dtype=np.int32)
            T.write_checkpoint(str(save_path), tensors)
            return str(save_path)

        def restore(self, sess, save_path):
            from .training import Adam
            for wf in self._wfs(sess):
                opt_state = wf.restore(str(save_path))
                st = self._step_of(wf)
                if st is not None and opt_state is not None:
                    if st.opt._adam is None:
                        st.opt._adam = Adam(st.opt.beta1, st.opt.beta2, st.opt.epsilon)
                    st.opt._adam.load_state(opt_state, list(wf.params))
                    if isinstance(st.global_step, Variable) and opt_state.get("global_step") is not None:
                        st.global_step.value = np.asarray(int(opt_state["global_step"]))


train = _Train()
compat = _Namespace(v1=_Namespace(logging=logging, Session=None, placeholder=None))


def is_gru_cell(cell):
    if cell is None or cell is CudnnCompatibleGRUCell:
        return True
    name = cell if isinstance(cell, str) else getattr(cell, "__name__", type(cell).__name__)
    return "GRU" in str(name).upper()


compat.v1.Session, compat.v1.placeholder = Session, placeholder
complex128 = np.complex128
abs = lambda x, *a, **k: Expr("abs", x)                            # noqa: E731,A001
reduce_max = lambda x, *a, **k: Expr("max", x)                     # noqa: E731
