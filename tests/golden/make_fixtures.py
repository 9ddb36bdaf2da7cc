#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own NumPy estimators in the build container.

Runs only where /root/reference exists (never on the GPU box).  The reference's
``Training*.py`` modules ``import tensorflow`` at the top, but the four estimator functions
used here (``Ising_local_energies``, ``Ising2D_local_energies`` x2, ``J1J2MatrixElements``,
``J1J2Slices``) are NumPy-only bodies.  TensorFlow is not installable here, so the import
statement is satisfied with an inert ``unittest.mock.MagicMock`` (recipe recorded in
SURVEY.md 8c); no TensorFlow arithmetic is replaced - the RNN half of the path is driven
through a duck-typed ``sess.run`` that returns log-probabilities computed by ``oracle/``.

The fixtures hold arrays only (inputs, weights, expected outputs): no reference source,
no bytecode.
"""
import importlib.util
import os
import sys
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import models  # noqa: E402
from rnnwavefunctions_amd import params as P  # noqa: E402


def load_reference_module(folder, filename):
    sys.modules["tensorflow"] = mock.MagicMock()
    for m in ("RNNwavefunction", "ComplexRNNwavefunction", "MDRNNcell"):
        sys.modules.pop(m, None)
    d = os.path.join(REF, folder)
    sys.path.insert(0, d)
    try:
        spec = importlib.util.spec_from_file_location("ref_" + folder.replace("/", "_"), os.path.join(d, filename))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(d)
    return mod


class FnSession:
    """Stands in for tf.Session: sess.run(tensor, feed_dict={ph: x}) -> fn(x)."""

    def __init__(self, fn):
        self.fn = fn
        self.calls = []

    def run(self, tensor, feed_dict=None):
        (x,) = feed_dict.values()
        self.calls.append(x.shape[0])
        return self.fn(x)


def g1_g2_j1j2(ref):
    out = {}
    rng = np.random.RandomState(7)
    case = 0
    for N in (4, 6, 8):
        if N <= 6:
            sig_all = ((np.arange(2 ** N)[:, None] >> np.arange(N)[::-1]) & 1).astype(np.int64)
        else:
            sig_all = rng.randint(0, 2, size=(24, N)).astype(np.int64)
        for J2v in (0.0, 0.2, 0.5):
            for periodic, marshall in ((False, False), (True, False), (False, True), (True, True)):
                J1 = np.ones(N)
                J2 = J2v * np.ones(N)
                Bz = 0.1 * np.arange(N)            # non-zero field to pin the (sigma-1/2).Bz term
                rows = np.full((len(sig_all), 2 * N + 2, N), -1, dtype=np.int32)
                elems = np.zeros((len(sig_all), 2 * N + 2), dtype=np.float32)
                nums = np.zeros(len(sig_all), dtype=np.int64)
                for k, sig in enumerate(sig_all):
                    sigmaH = np.zeros((2 * N + 2, N), dtype=np.int32)
                    me = np.zeros(2 * N + 2, dtype=np.float32)
                    num = ref.J1J2MatrixElements(J1, J2, Bz, sig, sigmaH, me, periodic, marshall)
                    nums[k] = num
                    rows[k, :num] = sigmaH[:num]
                    elems[k, :num] = me[:num]
                pre = "g1_%d_" % case
                out[pre + "meta"] = np.array([N, J2v, int(periodic), int(marshall)], dtype=np.float64)
                out[pre + "sigma"] = sig_all
                out[pre + "rows"] = rows
                out[pre + "elems"] = elems
                out[pre + "num"] = nums
                case += 1
    out["g1_ncases"] = np.array(case)

    # G2: J1J2Slices on a batch, both values of the Marshall_sign argument (which the reference
    # passes into the `periodic` positional slot, J1J2/TrainingRNN_J1J2.py:118)
    N, ns = 8, 32
    samples = np.stack([rng.permutation(np.repeat([0, 1], N // 2)) for _ in range(ns)]).astype(np.int32)
    J1, J2, Bz = np.ones(N), 0.5 * np.ones(N), np.zeros(N)
    for flag in (False, True):
        sigmas = np.zeros(((2 * N + 2) * ns, N), dtype=np.int32)
        H = np.zeros((2 * N + 2) * ns, dtype=np.float32)
        sigmaH = np.zeros((2 * N + 2, N), dtype=np.int32)
        me = np.zeros(2 * N + 2, dtype=np.float32)
        slices, total = ref.J1J2Slices(J1, J2, Bz, samples, sigmas, H, sigmaH, me, flag)
        pre = "g2_%d_" % int(flag)
        out[pre + "samples"] = samples
        out[pre + "sigmas"] = sigmas[:total].copy()
        out[pre + "H"] = H[:total].copy()
        out[pre + "offsets"] = np.array([s.start for s in slices] + [total], dtype=np.int64)
    return out


def g3_product_state(ref1d):
    """Ising_local_energies driven by a closed-form product state p(up)=0.3 per site."""
    samples = np.random.RandomState(0).randint(0, 2, (4, 5)).astype(np.int32)
    ns, N = samples.shape
    Jz = np.ones(N)
    Bx = 1.0

    def logp(x):
        return np.where(x == 1, np.log(0.3), np.log(0.7)).sum(axis=1)

    queue = np.zeros((N + 1, ns, N), dtype=np.int32)
    lp = np.zeros((N + 1) * ns, dtype=np.float64)
    e = ref1d.Ising_local_energies(Jz, Bx, samples, queue, None, "ph", lp, FnSession(logp))
    return {"g3_samples": samples, "g3_eloc": e}


def g4_estimators(ref1d, ref2d2d, ref2d1d, refj):
    out = {}
    rng = np.random.RandomState(11)

    # --- 1D TFIM, GRU pRNN f32, ns*(N+1) > 25000 so that the chunk loop runs twice
    N, nh, ns = 10, 12, 2400
    prm = P.randomize_biases(P.scale_kernels(P.init_gru_params([nh], seed=5), 2.0), 6)
    samples = rng.randint(0, 2, (ns, N)).astype(np.int32)
    Jz = 1.0 + 0.1 * rng.standard_normal(N)
    Bx = 0.9
    sess = FnSession(lambda x: models.prnn_log_probability(prm, x))
    queue = np.zeros((N + 1, ns, N), dtype=np.int32)
    lp = np.zeros((N + 1) * ns, dtype=np.float64)
    e = ref1d.Ising_local_energies(Jz, Bx, samples, queue, None, "ph", lp, sess)
    assert len(sess.calls) == 2, sess.calls
    for k, v in prm.items():
        out["g4a_param|" + k.replace("/", "|")] = v
    out.update(g4a_samples=samples, g4a_Jz=Jz, g4a_Bx=np.array(Bx), g4a_eloc=e, g4a_logp=lp.copy(),
               g4a_chunks=np.array(sess.calls))

    # --- same with Bx == 0 (the reference skips the flips but still evaluates zero rows, :42)
    queue[:] = 0
    lp[:] = 0
    e0 = ref1d.Ising_local_energies(Jz, 0.0, samples[:50], queue[:, :50].copy(), None, "ph",
                                    np.zeros((N + 1) * 50), FnSession(lambda x: models.prnn_log_probability(prm, x)))
    out["g4a_eloc_bx0"] = e0

    # --- 2D TFIM with the MDRNN (samples (ns, Nx, Ny)), f64
    Nx, Ny, nh, ns = 4, 4, 9, 96
    prm2 = P.scale_kernels(P.init_mdrnn_params(nh, seed=8), 1.5)
    samples2 = rng.randint(0, 2, (ns, Nx, Ny)).astype(np.int32)
    Jz2 = 1.0 + 0.1 * rng.standard_normal((Nx, Ny))
    Bx2 = 3.0
    sess = FnSession(lambda x: models.mdrnn_log_probability(prm2, x))
    queue = np.zeros((Nx * Ny + 1, ns, Nx, Ny), dtype=np.int32)
    lp = np.zeros((Nx * Ny + 1) * ns, dtype=np.float64)
    e2 = ref2d2d.Ising2D_local_energies(Jz2, Bx2, Nx, Ny, samples2, queue, None, "ph", lp, sess)
    for k, v in prm2.items():
        out["g4b_param|" + k.replace("/", "|")] = v
    out.update(g4b_samples=samples2, g4b_Jz=Jz2, g4b_Bx=np.array(Bx2), g4b_eloc=e2, g4b_logp=lp.copy())

    # --- 2D TFIM with the 1D GRU in f64 (samples (ns, Nx*Ny)), rectangular lattice
    Nx, Ny, nh, ns = 3, 4, 7, 64
    prm3 = P.randomize_biases(P.init_gru_params([nh], seed=9, dtype=np.float64), 10)
    samples3 = rng.randint(0, 2, (ns, Nx * Ny)).astype(np.int32)
    Jz3 = 1.0 + 0.1 * rng.standard_normal((Nx, Ny))
    Bx3 = 2.0
    sess = FnSession(lambda x: models.prnn_log_probability(prm3, x, dtype=np.float64))
    queue = np.zeros((Nx * Ny + 1, ns, Nx * Ny), dtype=np.int32)
    lp = np.zeros((Nx * Ny + 1) * ns, dtype=np.float64)
    e3 = ref2d1d.Ising2D_local_energies(Jz3, Bx3, Nx, Ny, samples3, queue, None, "ph", lp, sess)
    for k, v in prm3.items():
        out["g4c_param|" + k.replace("/", "|")] = v
    out.update(g4c_samples=samples3, g4c_Jz=Jz3, g4c_Bx=np.array(Bx3), g4c_eloc=e3, g4c_logp=lp.copy(),
               g4c_shape=np.array([Nx, Ny]))

    # --- J1J2: reference J1J2Slices + the reference's E_loc expression (TrainingRNN_J1J2.py:277-279
    #     is inline in run_J1J2, so only the slices come from the reference function)
    N, nh, ns = 10, 11, 48
    prm4 = P.randomize_biases(P.scale_kernels(
        P.init_gru_params([nh], seed=12, heads=("wf_dense_ampl", "wf_dense_phase")), 2.0), 13)
    samples4 = np.stack([rng.permutation(np.repeat([0, 1], N // 2)) for _ in range(ns)]).astype(np.int32)
    J1, J2, Bz = np.ones(N), 0.2 * np.ones(N), np.zeros(N)
    sigmas = np.zeros((2 * N * ns, N), dtype=np.int32)
    H = np.zeros(2 * N * ns, dtype=np.float32)
    slices, total = refj.J1J2Slices(J1, J2, Bz, samples4, sigmas, H, np.zeros((2 * N, N), dtype=np.int32),
                                    np.zeros(2 * N, dtype=np.float32), False)
    la = models.crnn_log_amplitude(prm4, sigmas[:total])
    e4 = np.zeros(ns, dtype=np.complex64)
    for n, s in enumerate(slices):
        e4[n] = H[s].dot(np.exp(la[s] - la[s][0]))
    for k, v in prm4.items():
        out["g4d_param|" + k.replace("/", "|")] = v
    out.update(g4d_samples=samples4, g4d_J2=np.array(0.2), g4d_eloc=e4, g4d_logamp=la,
               g4d_offsets=np.array([s.start for s in slices] + [total], dtype=np.int64))
    return out


def g7_notebook_trajectories():
    """The training trajectories the reference's two tutorial notebooks RECORD as cell output (every 10th step of run_1DTFIM at
    N=10, 10 units, 200 samples, lr 5e-3, 1000 steps; of run_J1J2 at N=10, J2=0.2, 10 units, 200 samples, lr 5e-4, 3000 steps):
    numbers printed by the reference's own TF-1 run - data, read with `json`, nothing executed.  They pin what no in-container
    run can: how fast and how far the reference's optimisation goes from its seeded initial state."""
    import json
    import re
    out = {}
    nb = json.load(open(os.path.join(REF, "Tutorials", "J1J2", "Tutorial_1DJ1J2.ipynb")))
    cell = next(c for c in nb["cells"] if c["cell_type"] == "code" and "run_J1J2(" in "".join(c["source"]))
    text = "".join("".join(o.get("text", [])) for o in cell["outputs"] if "text" in o)
    rows = re.findall(r"mean\(E\): \(([-0-9.e]+)([+-][0-9.e]+)j\), var\(E\): ([0-9.e-]+), #samples 200, #Step (\d+)", text)
    out["j1j2_step"] = np.array([int(r[3]) for r in rows])
    out["j1j2_re"] = np.array([float(r[0]) for r in rows])
    out["j1j2_im"] = np.array([float(r[1]) for r in rows])
    out["j1j2_var"] = np.array([float(r[2]) for r in rows])
    nb = json.load(open(os.path.join(REF, "Tutorials", "1DTFIM", "Tutorial_1DTFIM.ipynb")))
    cell = next(c for c in nb["cells"] if c["cell_type"] == "code" and "run_1DTFIM(" in "".join(c["source"]))
    text = "".join("".join(o.get("text", [])) for o in cell["outputs"] if "text" in o)
    rows = re.findall(r"mean\(E\): ([-0-9.e]+), var\(E\): ([0-9.e-]+), #samples (\d+), #Step (\d+)", text)
    out["tfim_step"] = np.array([int(r[3]) for r in rows])
    out["tfim_e"] = np.array([float(r[0]) for r in rows])
    out["tfim_var"] = np.array([float(r[1]) for r in rows])
    assert len(out["j1j2_step"]) == 301 and len(out["tfim_step"]) == 101
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "notebook_trajectories.npz"), **g7_notebook_trajectories())
    if "--trajectories-only" in sys.argv:
        return
    ref1d = load_reference_module("1DTFIM", "TrainingRNN_1DTFIM.py")
    refj = load_reference_module("J1J2", "TrainingRNN_J1J2.py")
    ref2d2d = load_reference_module("2DTFIM_2DRNN", "Training2DRNN_2DTFIM.py")
    ref2d1d = load_reference_module("2DTFIM_1DRNN", "Training1DRNN_2DTFIM.py")
    np.savez_compressed(os.path.join(HERE, "j1j2_matrix_elements.npz"), **g1_g2_j1j2(refj))
    np.savez_compressed(os.path.join(HERE, "ising_product_state.npz"), **g3_product_state(ref1d))
    np.savez_compressed(os.path.join(HERE, "estimators_oracle_driven.npz"),
                        **g4_estimators(ref1d, ref2d2d, ref2d1d, refj))
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
