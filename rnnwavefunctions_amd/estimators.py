"""Local-energy estimators with the reference's signatures, backed by the fused HIP path.

    Ising_local_energies    <- 1DTFIM/TrainingRNN_1DTFIM.py:13-75
    Ising2D_local_energies  <- 2DTFIM_2DRNN/Training2DRNN_2DTFIM.py:13-83 and
                               2DTFIM_1DRNN/Training1DRNN_2DTFIM.py:13-81
    J1J2MatrixElements      <- J1J2/TrainingRNN_J1J2.py:12-93
    J1J2Slices              <- J1J2/TrainingRNN_J1J2.py:95-127
    J1J2_local_energies     <- the inline loop J1J2/TrainingRNN_J1J2.py:255-279

The caller-owned scratch arrays of the reference (`queue_samples`, `log_probs`, `sigmas`, `H`, ...)
are accepted.  In the default ``mode="fused"`` the device never materialises the queue of flipped
configurations: `log_probs` is still filled exactly as the reference fills it (row 0 = log P(s),
row i+1 = log P(s with spin i flipped)); `queue_samples` only receives the diagonal block
(`queue_samples[0] = samples`) unless ``materialize_queue=True``.  ``mode="reference"`` runs the
reference's formulation step by step (host-built queue, <=25000-row chunks through ``sess.run``),
each chunk scored by the HIP log-probability kernel - useful to cross-check the fused path.
"""
from math import ceil

import numpy as np

from .compat import EvalOp


def _native_of(log_tensor, what):
    if not isinstance(log_tensor, EvalOp):
        raise TypeError("%s: the log-probability tensor must come from an rnnwavefunctions_amd wave function "
                        "(got %r); there is no TensorFlow/CPU path" % (what, type(log_tensor)))
    return log_tensor.wf


def _flip_queue(samples_flat, queue):
    """queue[0] = samples, queue[k+1] = samples with site k flipped (queue viewed as (N+1, ns, N))."""
    N = samples_flat.shape[1]
    queue[...] = samples_flat[None]
    idx = np.arange(N)
    queue[idx + 1, :, idx] ^= 1


def _reference_mode(samples, queue_samples, log_tensor, placeholder, log_probs, sess, Bx, max_rows=25000):
    ns = samples.shape[0]
    N = int(np.prod(samples.shape[1:]))
    q = queue_samples.reshape(N + 1, ns, N)
    if Bx != 0:
        _flip_queue(samples.reshape(ns, N), q)
    else:
        q[0] = samples.reshape(ns, N)
    total = (N + 1) * ns
    steps = ceil(total / max_rows)
    rows = queue_samples.reshape((total,) + samples.shape[1:])
    for i in range(steps):
        lo = (i * total) // steps
        hi = ((i + 1) * total) // steps if i < steps - 1 else total
        log_probs[lo:hi] = sess.run(log_tensor, feed_dict={placeholder: rows[lo:hi]})
    return log_probs[:total].reshape(N + 1, ns)


def _tfim_common(Jz, Bx, samples, queue_samples, log_tensor, placeholder, log_probs, sess, mode,
                 materialize_queue, diag_fn, what):
    wf = _native_of(log_tensor, what)
    samples = np.asarray(samples)
    ns = samples.shape[0]
    N = int(np.prod(samples.shape[1:]))
    if log_probs is None:
        log_probs = np.zeros((N + 1) * ns, dtype=np.float64)
    if mode == "reference":
        if queue_samples is None:
            queue_samples = np.zeros((N + 1,) + samples.shape, dtype=np.int32)
        lp = _reference_mode(samples.astype(np.int32), queue_samples, log_tensor, placeholder, log_probs, sess, Bx)
        return diag_fn(samples) - Bx * np.exp(0.5 * lp[1:] - 0.5 * lp[0]).sum(axis=0)
    if mode != "fused":
        raise ValueError("mode must be 'fused' or 'reference'")
    if queue_samples is not None:
        if materialize_queue and Bx != 0:
            _flip_queue(samples.reshape(ns, N).astype(np.int32), queue_samples.reshape(N + 1, ns, N))
        else:
            queue_samples[0] = samples
    return wf._native.tfim_eloc(samples, Jz, Bx, log_probs=log_probs)


def Ising_local_energies(Jz, Bx, samples, queue_samples, log_probs_tensor, samples_placeholder, log_probs, sess,
                         mode="fused", materialize_queue=False):
    """Local energies of the open 1D transverse-field Ising chain for a batch of configurations.

    samples (numsamples, N) ints in {0,1}; Jz (N,); Bx float; returns float64 (numsamples,).
    """
    def diag(s):
        sign = np.where(s[:, :-1] == s[:, 1:], 1.0, -1.0)
        return -(sign * np.asarray(Jz, dtype=np.float64)[: s.shape[1] - 1]).sum(axis=1)
    return _tfim_common(Jz, Bx, samples, queue_samples, log_probs_tensor, samples_placeholder, log_probs, sess,
                        mode, materialize_queue, diag, "Ising_local_energies")


def Ising2D_local_energies(Jz, Bx, Nx, Ny, samples, queue_samples, log_probs_tensor, samples_placeholder, log_probs,
                           sess, mode="fused", materialize_queue=False):
    """Local energies of the open 2D transverse-field Ising model.

    samples (numsamples, Nx, Ny) for the 2D RNN, or (numsamples, Nx*Ny) for the 1D RNN run over the
    flattened lattice (bonds are taken on the C-order reshape, flips on the flat index).
    """
    Jz2 = np.asarray(Jz, dtype=np.float64).reshape(Nx, Ny)

    def diag(s):
        s3 = s.reshape(s.shape[0], Nx, Ny)
        e = -(np.where(s3[:, :-1] == s3[:, 1:], 1.0, -1.0) * Jz2[:-1]).sum(axis=(1, 2))
        e -= (np.where(s3[:, :, :-1] == s3[:, :, 1:], 1.0, -1.0) * Jz2[:, :-1]).sum(axis=(1, 2))
        return e
    return _tfim_common(Jz2, Bx, samples, queue_samples, log_probs_tensor, samples_placeholder, log_probs, sess,
                        mode, materialize_queue, diag, "Ising2D_local_energies")


# ---------------------------------------------------------------------------------------------------
# J1-J2 Heisenberg chain
# ---------------------------------------------------------------------------------------------------
def J1J2MatrixElements(J1, J2, Bz, sigmap, sigmaH, matrixelements, periodic=False, Marshall_sign=False):
    """Connected configurations and matrix elements of one configuration (host NumPy, as in the
    reference).  Fills ``sigmaH[:num]`` / ``matrixelements[:num]`` (row 0 = diagonal) and returns num.
    The fused device path (J1J2_local_energies) enumerates the same set on the GPU."""
    s = np.asarray(sigmap)
    N = len(Bz)
    J1 = np.asarray(J1, dtype=np.float64)
    J2 = np.asarray(J2, dtype=np.float64)
    diag = float(np.dot(s - 0.5, Bz))
    rows = [s]
    vals = []
    for dist, J in ((1, J1), (2, J2)):
        lim = N if periodic else N - dist
        i = np.arange(lim)
        j = (i + dist) % N
        live = J[:lim] != 0.0 if dist == 2 else np.ones(lim, dtype=bool)
        anti = s[i] != s[j]
        for term in (np.where(anti, -0.25, 0.25) * J[:lim])[live]:     # sequential, as the reference sums it
            diag += float(term)
        for a in i[anti & (J[:lim] != 0.0)]:
            b = (a + dist) % N
            t = s.copy()
            t[a], t[b] = s[b], s[a]
            rows.append(t)
            vals.append(-J[a] / 2 if (Marshall_sign and dist == 1) else J[a] / 2)
    num = len(rows)
    sigmaH[:num] = np.stack(rows)
    matrixelements[0] = diag
    matrixelements[1:num] = vals
    return num


def J1J2Slices(J1, J2, Bz, sigmasp, sigmas, H, sigmaH, matrixelements, Marshall_sign):
    """Ragged packing of J1J2MatrixElements over a batch; returns (list of slices, total length).

    Reference quirk kept on purpose: the reference forwards ``Marshall_sign`` into
    the ``periodic`` positional slot of J1J2MatrixElements (TrainingRNN_J1J2.py:118)."""
    slices, total = [], 0
    for sigmap in np.asarray(sigmasp):
        num = J1J2MatrixElements(J1, J2, Bz, sigmap, sigmaH, matrixelements, Marshall_sign)
        s = slice(total, total + num)
        H[s] = matrixelements[:num]
        sigmas[s] = sigmaH[:num]
        slices.append(s)
        total += num
    return slices, total


def J1J2_local_energies(J1, J2, Bz, samples, log_amps_tensor, periodic=False, Marshall_sign=False,
                        return_num_connected=False):
    """Fused device version of the reference's J1J2 step: connected configurations, log-amplitudes
    (with hidden-state prefix reuse) and  E_loc[n] = sum_k H_k exp(log psi(s'_k) - log psi(s))
    -> complex64 (numsamples,)."""
    wf = _native_of(log_amps_tensor, "J1J2_local_energies")
    e, ncon = wf._native.j1j2_eloc(samples, J1, J2, Bz, periodic, Marshall_sign)
    return (e, ncon) if return_num_connected else e
