"""NumPy restatement of the reference's local-energy estimators.  TEST INFRASTRUCTURE ONLY.

These follow the *reference formulation*: every connected configuration is
materialised and scored from site 0 by a caller-supplied ``log_prob_fn`` in chunks
of at most 25 000 (TFIM) / 30 000 (J1J2) rows.  They are pinned against the
reference's own NumPy code by tests/golden (G1-G4 of SURVEY.md 8c).
"""
from math import ceil

import numpy as np


def _chunks(total, max_rows):
    """Chunk boundaries of 1DTFIM/TrainingRNN_1DTFIM.py:56-64 (integer slicing of the
    range into ``ceil(total/max_rows)`` nearly equal pieces)."""
    steps = ceil(total / max_rows) if total > 0 else 0
    for i in range(steps):
        lo = (i * total) // steps
        hi = ((i + 1) * total) // steps if i < steps - 1 else total
        yield lo, hi


def _eval_chunked(configs, log_fn, max_rows, dtype):
    out = np.zeros(configs.shape[0], dtype=dtype)
    for lo, hi in _chunks(configs.shape[0], max_rows):
        out[lo:hi] = log_fn(configs[lo:hi])
    return out


def _bond_sign(a, b):
    """+1 for aligned spins, -1 for anti-aligned (1DTFIM/TrainingRNN_1DTFIM.py:32-36)."""
    return np.where(a == b, 1.0, -1.0)


def ising_local_energies(Jz, Bx, samples, log_prob_fn, return_log_probs=False):
    """1DTFIM/TrainingRNN_1DTFIM.py:13-75.
    E_loc = -sum_i Jz[i] s_i s_{i+1} - Bx sum_i exp(0.5 (logP(flip_i s) - logP(s)))."""
    samples = np.asarray(samples)
    ns, N = samples.shape
    eloc = np.zeros(ns, dtype=np.float64)
    for i in range(N - 1):                                                 # :31-38
        eloc += _bond_sign(samples[:, i], samples[:, i + 1]) * (-Jz[i])
    queue = np.zeros((N + 1, ns, N), dtype=np.int32)
    queue[0] = samples                                                     # :40
    if Bx != 0:                                                            # :42
        for i in range(N):                                                 # :43-48
            queue[i + 1] = samples
            queue[i + 1][:, i] = 1 - samples[:, i]
    lp = _eval_chunked(queue.reshape((N + 1) * ns, N), log_prob_fn, 25000, np.float64)  # :56-65
    lp = lp.reshape(N + 1, ns)                                             # :70
    eloc += -Bx * np.exp(0.5 * lp[1:] - 0.5 * lp[0]).sum(axis=0)           # :74
    return (eloc, lp) if return_log_probs else eloc


def ising2d_local_energies(Jz, Bx, Nx, Ny, samples, log_prob_fn, return_log_probs=False):
    """2DTFIM_2DRNN/Training2DRNN_2DTFIM.py:13-83 for samples (ns, Nx, Ny), and
    2DTFIM_1DRNN/Training1DRNN_2DTFIM.py:13-81 for samples (ns, Nx*Ny) (bonds taken on the
    C-order reshape, :27; flips on the flat index, :55-60)."""
    samples = np.asarray(samples)
    ns = samples.shape[0]
    flat = samples.ndim == 2
    s3 = samples.reshape(ns, Nx, Ny)
    N = Nx * Ny
    eloc = np.zeros(ns, dtype=np.float64)
    for i in range(Nx - 1):                                                # :33-40
        eloc += (_bond_sign(s3[:, i], s3[:, i + 1]) * (-Jz[i, :])).sum(axis=1)
    for i in range(Ny - 1):                                                # :42-49
        eloc += (_bond_sign(s3[:, :, i], s3[:, :, i + 1]) * (-Jz[:, i])).sum(axis=1)
    sflat = samples.reshape(ns, N)
    queue = np.zeros((N + 1, ns, N), dtype=np.int32)
    queue[0] = sflat                                                       # :52
    if Bx != 0:
        for k in range(N):                                                 # :54-61 (k = i*Ny + j)
            queue[k + 1] = sflat
            queue[k + 1][:, k] = 1 - sflat[:, k]
    shape = ((N + 1) * ns, N) if flat else ((N + 1) * ns, Nx, Ny)
    lp = _eval_chunked(queue.reshape(shape), log_prob_fn, 25000, np.float64)   # :66-75
    lp = lp.reshape(N + 1, ns)
    eloc += -Bx * np.exp(0.5 * lp[1:] - 0.5 * lp[0]).sum(axis=0)           # :81
    return (eloc, lp) if return_log_probs else eloc


def j1j2_matrix_elements(J1, J2, Bz, sigmap, periodic=False, Marshall_sign=False):
    """J1J2/TrainingRNN_J1J2.py:12-93 for one configuration.
    Returns (sigmaH (num, N) int32, matrixelements (num,) float32): row 0 is the diagonal."""
    sigmap = np.asarray(sigmap)
    N = len(Bz)
    diag = np.dot(sigmap - 0.5, Bz)                                        # :32
    lim1 = N if periodic else N - 1                                        # :36-39
    lim2 = N if periodic else N - 2                                        # :41-44
    for s in range(lim1):                                                  # :46-50
        diag += 0.25 * J1[s] * (1.0 if sigmap[s] == sigmap[(s + 1) % N] else -1.0)
    for s in range(lim2):                                                  # :52-57
        if J2[s] != 0.0:
            diag += 0.25 * J2[s] * (1.0 if sigmap[s] == sigmap[(s + 2) % N] else -1.0)
    rows = [sigmap.astype(np.int32)]
    elems = [diag]
    for dist, J, lim in ((1, J1, lim1), (2, J2, lim2)):                    # :68-92
        for s in range(lim):
            t = (s + dist) % N
            if J[s] != 0.0 and sigmap[s] != sigmap[t]:
                sig = sigmap.astype(np.int32).copy()
                sig[s], sig[t] = sigmap[t], sigmap[s]
                rows.append(sig)
                if dist == 1 and Marshall_sign:
                    elems.append(-J[s] / 2)                                # :76-77
                else:
                    elems.append(+J[s] / 2)                                # :79, :90
    return np.stack(rows), np.asarray(elems, dtype=np.float32)


def j1j2_slices(J1, J2, Bz, sigmasp, periodic=False, Marshall_sign=False):
    """J1J2/TrainingRNN_J1J2.py:95-127 (ragged packing).  Returns
    (sigmas (total, N) int32, H (total,) float32, offsets (ns+1,) int64)."""
    all_rows, all_h, offs = [], [], [0]
    for sig in np.asarray(sigmasp):
        r, h = j1j2_matrix_elements(J1, J2, Bz, sig, periodic, Marshall_sign)
        all_rows.append(r)
        all_h.append(h)
        offs.append(offs[-1] + len(h))
    return np.concatenate(all_rows), np.concatenate(all_h), np.asarray(offs, dtype=np.int64)


def j1j2_local_energies(J1, J2, Bz, samples, log_amp_fn, periodic=False, Marshall_sign=False):
    """J1J2/TrainingRNN_J1J2.py:255-279 -> complex64 (ns,)."""
    sigmas, H, offs = j1j2_slices(J1, J2, Bz, samples, periodic, Marshall_sign)
    la = _eval_chunked(sigmas, log_amp_fn, 30000, np.complex64)            # :260-270
    eloc = np.zeros(len(offs) - 1, dtype=np.complex64)
    for n in range(len(offs) - 1):                                         # :276-279
        s = slice(offs[n], offs[n + 1])
        eloc[n] = H[s].dot(np.exp(la[s] - la[s][0]))
    return eloc


def energy_moments(eloc):
    """1DTFIM/TrainingRNN_1DTFIM.py:206-207 and J1J2/TrainingRNN_J1J2.py:281-282:
    mean (complex for J1J2) and population variance of the real part."""
    eloc = np.asarray(eloc)
    return np.mean(eloc), np.var(np.real(eloc))
