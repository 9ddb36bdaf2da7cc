"""NumPy restatement of the reference's RNN wave functions.  TEST INFRASTRUCTURE ONLY.

Every function cites the reference lines it follows (paths relative to
/root/reference).  TensorFlow-1.13.1 op semantics that are not in the reference
tree (GRU cell, Dense, softmax, multinomial, l2_normalize, elu) are restated from
the published TF source; SURVEY.md 8c lists them and what pins each.

Parameters are plain dicts keyed by the TensorFlow variable names the reference's
graph would create, e.g.
``RNNwavefunction/multi_rnn_cell/cell_0/cudnn_compatible_gru_cell/gates/kernel``.
"""
import numpy as np

GRU = "multi_rnn_cell/cell_%d/cudnn_compatible_gru_cell/"


def _p(params, scope, name):
    return params[scope + "/" + name]


def num_gru_layers(params, scope="RNNwavefunction"):
    n = 0
    while (scope + "/" + GRU % n + "gates/kernel") in params:
        n += 1
    return n


def sigmoid(x):
    one = x.dtype.type(1)
    return one / (one + np.exp(-x))


def softmax(z):
    # tf.nn.softmax: exp(z - max) / sum
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def gru_cell(x, h, params, scope, layer):
    """tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell.call (TF 1.13.1), invoked at
    1DTFIM/RNNwavefunction.py:66,108.  Gate columns are [r | u]; separate input and
    hidden candidate projections, each with its own bias."""
    pre = GRU % layer
    Wg = _p(params, scope, pre + "gates/kernel")
    bg = _p(params, scope, pre + "gates/bias")
    Wci = _p(params, scope, pre + "candidate/input_projection/kernel")
    bci = _p(params, scope, pre + "candidate/input_projection/bias")
    Wch = _p(params, scope, pre + "candidate/hidden_projection/kernel")
    bch = _p(params, scope, pre + "candidate/hidden_projection/bias")
    nh = h.shape[1]
    g = sigmoid(np.concatenate([x, h], axis=1) @ Wg + bg)
    r, u = g[:, :nh], g[:, nh:]
    c = np.tanh((x @ Wci + bci) + r * (h @ Wch + bch))
    one = h.dtype.type(1)
    return (one - u) * c + u * h


def multi_gru(x, states, params, scope):
    """tf.nn.rnn_cell.MultiRNNCell: layer l feeds layer l+1 (1DTFIM/RNNwavefunction.py:32)."""
    new_states = []
    for layer, h in enumerate(states):
        x = gru_cell(x, h, params, scope, layer)
        new_states.append(x)
    return x, new_states


def _zero_states(params, scope, batch, dtype):
    out = []
    for layer in range(num_gru_layers(params, scope)):
        nh = _p(params, scope, GRU % layer + "candidate/hidden_projection/kernel").shape[0]
        out.append(np.zeros((batch, nh), dtype=dtype))
    return out


def _one_hot(col, dtype):
    return np.eye(2, dtype=dtype)[col]


def multinomial_2(logits, u):
    """tf.multinomial / tf.random.categorical, CPU kernel (TF 1.13.1
    core/kernels/multinomial_op.cc): un-normalised CDF of exp(logit - max) in double,
    non-finite logits skipped, index = upper_bound(cdf, u * total).  ``u`` takes the
    place of TF's Philox uniform double."""
    lg = logits.astype(np.float64)
    finite = np.isfinite(logits)
    mx = np.max(np.where(finite, lg, -np.inf), axis=1, keepdims=True)
    with np.errstate(invalid="ignore"):
        e = np.where(finite, np.exp(lg - mx), 0.0)
    cdf = np.cumsum(e, axis=1)
    to_find = u * cdf[:, -1]
    idx = (cdf <= to_find[:, None]).sum(axis=1)
    return np.minimum(idx, logits.shape[1] - 1).astype(np.int64)


# ----------------------------------------------------------------------------------------------
# 1D positive RNN wave function (1DTFIM/RNNwavefunction.py, 2DTFIM_1DRNN/RNNwavefunction.py)
# ----------------------------------------------------------------------------------------------

def prnn_site_probs(params, samples, scope="RNNwavefunction", dtype=np.float32):
    """Teacher-forced conditionals p_n(.|sigma_<n): 1DTFIM/RNNwavefunction.py:97-111.
    Returns (B, N, 2) in the cell dtype."""
    samples = np.asarray(samples)
    B, N = samples.shape
    Wd = _p(params, scope, "wf_dense/kernel")
    bd = _p(params, scope, "wf_dense/bias")
    x = np.zeros((B, 2), dtype=dtype)                      # :97-100 first input is the zero vector
    states = _zero_states(params, scope, B, dtype)          # :105
    probs = np.empty((B, N, 2), dtype=dtype)
    for n in range(N):                                      # :107
        out, states = multi_gru(x, states, params, scope)   # :108
        probs[:, n] = softmax(out @ Wd + bd)                # :109
        x = _one_hot(samples[:, n], dtype)                  # :111
    return probs


def prnn_log_probability(params, samples, scope="RNNwavefunction", dtype=np.float32):
    """1DTFIM/RNNwavefunction.py:113-116: probs cast to f64, select, log, sum -> f64 (B,)."""
    probs = prnn_site_probs(params, samples, scope, dtype).astype(np.float64)
    sel = np.take_along_axis(probs, np.asarray(samples)[:, :, None].astype(np.int64), axis=2)[:, :, 0]
    with np.errstate(divide="ignore"):
        return np.log(sel).sum(axis=1)


def prnn_paritysym_log_probability(params, samples, scope="RNNwavefunction", dtype=np.float32):
    """1DTFIM/RNNwavefunction_paritysym.py:122-145: log(0.5 (P(s) + P(reversed s)))."""
    samples = np.asarray(samples)
    lp1 = prnn_log_probability(params, samples, scope, dtype)
    lp2 = prnn_log_probability(params, samples[:, ::-1], scope, dtype)      # :125
    with np.errstate(divide="ignore"):
        return np.log(0.5 * (np.exp(lp1) + np.exp(lp2)))                    # :145


def prnn_sample(params, N, u, scope="RNNwavefunction", dtype=np.float32):
    """Ancestral sampling, 1DTFIM/RNNwavefunction.py:52-72, with explicit uniforms
    ``u`` (numsamples, N) in place of TF's Philox.  Returns (samples int64, log_prob f64)."""
    ns = u.shape[0]
    Wd = _p(params, scope, "wf_dense/kernel")
    bd = _p(params, scope, "wf_dense/bias")
    x = np.zeros((ns, 2), dtype=dtype)                      # :52-55
    states = _zero_states(params, scope, ns, dtype)         # :62
    samples = np.empty((ns, N), dtype=np.int64)
    logp = np.zeros(ns, dtype=np.float64)
    for n in range(N):                                      # :65
        out, states = multi_gru(x, states, params, scope)   # :66
        p = softmax(out @ Wd + bd)                          # :67
        with np.errstate(divide="ignore"):
            s = multinomial_2(np.log(p), u[:, n])           # :68
        samples[:, n] = s
        logp += np.log(p[np.arange(ns), s].astype(np.float64))
        x = _one_hot(s, dtype)                              # :70
    return samples, logp


# ----------------------------------------------------------------------------------------------
# 1D complex RNN wave function with U(1) mask (J1J2/ComplexRNNwavefunction.py)
# ----------------------------------------------------------------------------------------------

def _heavyside(x):
    """J1J2/ComplexRNNwavefunction.py:11-13 : 1 where x >= 0 else 0."""
    return (0.5 * (np.sign(np.sign(x) + x.dtype.type(0.1)) + 1.0)).astype(x.dtype)


def _l2_normalize(x, eps=1e-30):
    """tf.nn.l2_normalize(axis=1): x * rsqrt(max(sum x^2, eps))."""
    ss = np.maximum((x * x).sum(axis=1, keepdims=True), x.dtype.type(eps))
    return (x / np.sqrt(ss)).astype(x.dtype)


def _crnn_masked_ampl(out, Wa, ba, n, N, num_up):
    ampl = np.sqrt(softmax(out @ Wa + ba))                  # :5-6, :83 / :143
    if n >= N / 2:                                          # :85 / :147
        baseline = num_up.dtype.type(N // 2 - 1)
        num_down = num_up.dtype.type(n) - num_up
        act_up = _heavyside(baseline - num_up)              # :89 / :151
        act_down = _heavyside(baseline - num_down)
        ampl = ampl * np.stack([act_down, act_up], axis=1)  # :92 / :154
        ampl = _l2_normalize(ampl)                          # :93 / :155
    return ampl


def crnn_sample(params, N, u, scope="RNNwavefunction"):
    """J1J2/ComplexRNNwavefunction.py:63-101 with explicit uniforms."""
    dtype = np.float32
    ns = u.shape[0]
    Wa = _p(params, scope, "wf_dense_ampl/kernel")
    ba = _p(params, scope, "wf_dense_ampl/bias")
    x = np.zeros((ns, 2), dtype=dtype)
    states = _zero_states(params, scope, ns, dtype)
    samples = np.empty((ns, N), dtype=np.int64)
    num_up = np.zeros(ns, dtype=np.float32)
    for n in range(N):
        out, states = multi_gru(x, states, params, scope)
        ampl = _crnn_masked_ampl(out, Wa, ba, n, N, num_up)
        with np.errstate(divide="ignore"):
            s = multinomial_2(np.log(ampl ** 2), u[:, n])   # :95
        samples[:, n] = s
        num_up += s.astype(np.float32)
        x = _one_hot(s, dtype)
    return samples


def crnn_log_amplitude(params, samples, scope="RNNwavefunction", dtype=np.float32):
    """J1J2/ComplexRNNwavefunction.py:126-167 -> complex64 (B,)  (dtype=float64 -> complex128: used only to take
    finite differences of the cost in the gradient tests)."""
    ctype = np.complex64 if dtype == np.float32 else np.complex128
    samples = np.asarray(samples)
    B, N = samples.shape
    Wa = _p(params, scope, "wf_dense_ampl/kernel")
    ba = _p(params, scope, "wf_dense_ampl/bias")
    Wp = _p(params, scope, "wf_dense_phase/kernel")
    bp = _p(params, scope, "wf_dense_phase/bias")
    x = np.zeros((B, 2), dtype=dtype)
    states = _zero_states(params, scope, B, dtype)
    sel = np.empty((B, N), dtype=ctype)
    rows = np.arange(B)
    for n in range(N):
        out, states = multi_gru(x, states, params, scope)
        num_up = samples[:, :n].sum(axis=1).astype(dtype)               # :148
        ampl = _crnn_masked_ampl(out, Wa, ba, n, N, num_up)
        z = out @ Wp + bp
        phase = (dtype(np.pi) * (z / (dtype(1) + np.abs(z)))).astype(dtype)   # :8-9, :145
        amp_c = ampl.astype(ctype) * np.exp(1j * phase.astype(ctype)).astype(ctype)  # :157
        sel[:, n] = amp_c[rows, samples[:, n]]                            # :165-167 one-hot select
        x = _one_hot(samples[:, n], dtype)                                # :161
    with np.errstate(divide="ignore"):
        return np.log(sel).sum(axis=1).astype(ctype)                      # :167


# ----------------------------------------------------------------------------------------------
# 2D MDRNN wave function (2DTFIM_2DRNN/MDRNNcell.py, 2DTFIM_2DRNN/RNNwavefunction.py), float64
# ----------------------------------------------------------------------------------------------

def mdrnn_cell(xh, xv, hh, hv, params, scope, name="rnn_0"):
    """2DTFIM_2DRNN/MDRNNcell.py:51-66: elu(xh Uh + hh Wh + xv Uv + hv Wv + b)."""
    pre = (xh @ _p(params, scope, "Uh_" + name) + hh @ _p(params, scope, "Wh_" + name)
           + xv @ _p(params, scope, "Uv_" + name) + hv @ _p(params, scope, "Wv_" + name)
           + _p(params, scope, "b_" + name))                              # :54-60
    return np.where(pre > 0, pre, np.expm1(np.minimum(pre, 0.0)))          # :62 tf.nn.elu


def zigzag_order(Nx, Ny):
    """Visit order of 2DTFIM_2DRNN/RNNwavefunction.py:90-113: list of (nx, ny, nx_horizontal_neighbour)."""
    order = []
    for ny in range(Ny):
        xs = range(Nx) if ny % 2 == 0 else range(Nx - 1, -1, -1)
        for nx in xs:
            order.append((nx, ny, nx - 1 if ny % 2 == 0 else nx + 1))
    return order


def _mdrnn_run(params, Nx, Ny, B, scope, samples=None, u=None):
    dtype = np.float64
    nh = _p(params, scope, "Wh_rnn_0").shape[0]
    Wd = _p(params, scope, "wf_dense/kernel")
    bd = _p(params, scope, "wf_dense/bias")
    zeros_h = np.zeros((B, nh), dtype=dtype)
    zeros_x = np.zeros((B, 2), dtype=dtype)
    h, x = {}, {}                                           # tuple keys; see SURVEY.md 2.2-3
    out_samples = np.empty((B, Nx, Ny), dtype=np.int64)
    logp = np.zeros(B, dtype=np.float64)
    rows = np.arange(B)
    k = 0
    for nx, ny, nxh in zigzag_order(Nx, Ny):
        hh, xh = h.get((nxh, ny), zeros_h), x.get((nxh, ny), zeros_x)     # :70-81 zero boundaries
        hv, xv = h.get((nx, ny - 1), zeros_h), x.get((nx, ny - 1), zeros_x)  # :84-87
        hn = mdrnn_cell(xh, xv, hh, hv, params, scope)                    # :96 / :108
        p = softmax(hn @ Wd + bd)                                          # :98
        if samples is None:
            with np.errstate(divide="ignore"):
                s = multinomial_2(np.log(p), u[:, k])                      # :99
        else:
            s = samples[:, nx, ny]                                         # :182
        out_samples[:, nx, ny] = s                                         # :100, :116
        with np.errstate(divide="ignore"):
            logp += np.log(p[rows, s])                                     # :195-198
        h[(nx, ny)] = hn
        x[(nx, ny)] = _one_hot(s, dtype)                                   # :101 / :182
        k += 1
    return out_samples, logp


def mdrnn_log_probability(params, samples, scope="RNNwavefunction"):
    """2DTFIM_2DRNN/RNNwavefunction.py:120-200; samples (B, Nx, Ny) indexed [b, nx, ny]."""
    samples = np.asarray(samples)
    B, Nx, Ny = samples.shape
    return _mdrnn_run(params, Nx, Ny, B, scope, samples=samples)[1]


def mdrnn_sample(params, Nx, Ny, u, scope="RNNwavefunction"):
    """2DTFIM_2DRNN/RNNwavefunction.py:35-118; ``u`` (numsamples, Nx*Ny), column k is the
    k-th visited site of the zig-zag path.  Returns (samples (ns,Nx,Ny) int64, log_prob)."""
    return _mdrnn_run(params, Nx, Ny, u.shape[0], scope, u=u)
