"""GPU parity tests of the 2D MDRNN path (float64 end to end, as the reference: 2DTFIM_2DRNN/).

Tolerances: log P |hip - oracle| <= 1e-11 * N (f64; only the summation order differs),
            E_loc relative 1e-10, samples identical.
"""
import numpy as np
import pytest

from conftest import all_configs, golden_params
from oracle import estimators as E
from oracle import models as M
from oracle import philox
from rnnwavefunctions_amd import params as P

pytestmark = pytest.mark.gpu


def make_wf(Nx, Ny, H, prm):
    from rnnwavefunctions_amd import _lib
    wf = _lib.NativeWavefunction(_lib.MODEL_MDRNN2D, Nx, Ny, (H,))
    wf.set_params(prm, scope="RNNwavefunction")
    return wf


@pytest.mark.parametrize("Nx,Ny,H,B", [(4, 4, 9, 40), (3, 5, 20, 33), (6, 6, 50, 50), (12, 12, 50, 20), (5, 4, 36, 17),
                                        (4, 3, 64, 19), (4, 3, 68, 19), (3, 4, 69, 17), (4, 4, 84, 20)])      # 69..84 units: the five-tile layout
def test_log_prob_matches_oracle(Nx, Ny, H, B):
    prm = P.scale_kernels(P.init_mdrnn_params(H, seed=H), 1.5 if Nx * Ny <= 64 else 1.0)
    wf = make_wf(Nx, Ny, H, prm)
    s = np.random.RandomState(Nx * Ny).randint(0, 2, (B, Nx, Ny)).astype(np.int32)
    got = wf.log_prob(s)
    ref = M.mdrnn_log_probability(prm, s)
    print("%dx%d H=%d: |hip-oracle|=%.2e" % (Nx, Ny, H, np.abs(got - ref).max()))
    assert np.all(np.isfinite(ref))
    assert np.allclose(got, ref, rtol=0, atol=1e-11 * Nx * Ny)


def test_normalisation_3x3():
    prm = P.scale_kernels(P.init_mdrnn_params(12, seed=2), 2.0)
    wf = make_wf(3, 3, 12, prm)
    lp = wf.log_prob(all_configs(9).reshape(-1, 3, 3))
    assert abs(np.exp(lp).sum() - 1) < 1e-12


def test_tfim2d_eloc_matches_reference_golden(golden_estimators):
    """G4b: the reference's Ising2D_local_energies (2DTFIM_2DRNN) driven by the oracle."""
    g = golden_estimators
    prm = golden_params(g, "g4b")
    wf = make_wf(4, 4, 9, prm)
    s = g["g4b_samples"]
    lp = np.zeros(17 * s.shape[0])
    e = wf.tfim_eloc(s, g["g4b_Jz"], float(g["g4b_Bx"]), log_probs=lp)
    print("G4b: max|lp diff|=%.2e" % np.abs(lp - g["g4b_logp"]).max())
    assert np.allclose(lp, g["g4b_logp"], rtol=0, atol=1e-10)
    assert np.allclose(e, g["g4b_eloc"], rtol=1e-10)


@pytest.mark.parametrize("Nx,Ny,H,ns", [(5, 3, 20, 37), (6, 6, 50, 16), (3, 6, 9, 70), (4, 3, 72, 21), (3, 3, 84, 17)])
def test_tfim2d_eloc_fused_equals_reference_formulation(Nx, Ny, H, ns):
    prm = P.scale_kernels(P.init_mdrnn_params(H, seed=7), 1.5)
    wf = make_wf(Nx, Ny, H, prm)
    rng = np.random.RandomState(1)
    s = rng.randint(0, 2, (ns, Nx, Ny)).astype(np.int32)
    Jz = 1.0 + 0.1 * rng.standard_normal((Nx, Ny))
    lp = np.zeros((Nx * Ny + 1) * ns)
    e = wf.tfim_eloc(s, Jz, 3.0, log_probs=lp)
    e_ref, lp_ref = E.ising2d_local_energies(Jz, 3.0, Nx, Ny, s, lambda x: M.mdrnn_log_probability(prm, x),
                                             return_log_probs=True)
    assert np.allclose(lp, lp_ref.ravel(), rtol=0, atol=1e-11 * Nx * Ny)
    assert np.allclose(e, e_ref, rtol=1e-10)


def test_sampling_matches_oracle_stream():
    Nx, Ny, H, ns = 6, 5, 20, 500
    prm = P.scale_kernels(P.init_mdrnn_params(H, seed=3), 2.0)
    wf = make_wf(Nx, Ny, H, prm)
    s, lg = wf.sample(ns, seed=21, step=4, return_log=True)
    s_ref, lg_ref = M.mdrnn_sample(prm, Nx, Ny, philox.uniforms(21, 4, 0, ns, Nx * Ny))
    assert s.shape == (ns, Nx, Ny)
    assert np.array_equal(s, s_ref)
    assert np.allclose(lg, lg_ref, atol=1e-10)
    assert np.array_equal(np.concatenate([wf.sample(123, 21, 4, 0), wf.sample(377, 21, 4, 123)]), s)


def test_vmc_step_2d_and_facade():
    from rnnwavefunctions_amd import compat as tf
    from rnnwavefunctions_amd.TFIM2D_2DRNN.Training2DRNN_2DTFIM import Ising2D_local_energies, MDRNNcell, RNNwavefunction
    Nx, Ny, numsamples, Bx = 4, 5, 100, 3.0
    wf = RNNwavefunction(Nx, Ny, units=[20], cell=MDRNNcell, seed=111)
    assert wf.num_params() == 2 * 400 + 2 * 2 * 20 + 20 + 42
    sess = tf.Session(graph=wf.graph)
    samples_ = wf.sample(numsamples=numsamples, inputdim=2)
    ph = tf.placeholder(dtype=tf.int32, shape=(None, Nx, Ny))
    t = wf.log_probability(ph, inputdim=2)
    Jz = +np.ones((Nx, Ny))
    queue = np.zeros((Nx * Ny + 1, numsamples, Nx, Ny), dtype=np.int32)
    log_probs = np.zeros((Nx * Ny + 1) * numsamples)
    samples = sess.run(samples_)
    e = Ising2D_local_energies(Jz, Bx, Nx, Ny, samples, queue, t, ph, log_probs, sess)
    prm = wf.get_params()
    e_ref = E.ising2d_local_energies(Jz, Bx, Nx, Ny, samples, lambda x: M.mdrnn_log_probability(prm, x))
    assert np.allclose(e, e_ref, rtol=1e-10)
    e2 = Ising2D_local_energies(Jz, Bx, Nx, Ny, samples, queue, t, ph, np.zeros_like(log_probs), sess, mode="reference")
    assert np.allclose(e2, e, rtol=1e-10)
    out = wf._native.vmc_step(numsamples, seed=111, step=0, couplings=np.append(Jz.ravel(), Bx), want_samples=True,
                              want_eloc=True)
    assert np.array_equal(out["samples"], samples)
    assert np.allclose(out["eloc"], e, rtol=1e-12)
    assert np.isclose(out["moments"][0] / numsamples, e.mean(), rtol=1e-12)


def test_config4_at_full_size():
    """BASELINE config 4 exactly as stated (2DTFIM_2DRNN/run_2dTFIM.py:10: 12x12, num_units=50, numsamples=10000;
    the 600 MB state buffer and the full persistent grid): local energies and energy per site of HIP vs the float64 oracle on
    512 samples of the same sample matrix (145 x 512 lattices of 144 sites), plus size-independent properties on the whole batch."""
    Nx = Ny = 12
    H, ns = 50, 10000
    prm = P.init_mdrnn_params(H, seed=111)
    wf = make_wf(Nx, Ny, H, prm)
    Jz = np.ones((Nx, Ny))
    out = wf.vmc_step(ns, seed=111, step=0, couplings=np.append(Jz.ravel(), 3.0), want_samples=True, want_eloc=True)
    s, e = out["samples"], out["eloc"]
    assert np.all(np.isfinite(e))
    sub = np.arange(0, ns, ns // 512)[:512]
    e_ref = np.concatenate([E.ising2d_local_energies(Jz, 3.0, Nx, Ny, s[sub[k:k + 64]], lambda x: M.mdrnn_log_probability(prm, x))
                            for k in range(0, 512, 64)])
    per_site = np.abs(e[sub] - e_ref).max() / (Nx * Ny)
    d_mean = abs(e[sub].mean() - e_ref.mean()) / (Nx * Ny)
    print("cfg4: over 512 samples max |E_loc diff| / N = %.2e, |<E> diff| / N = %.2e" % (per_site, d_mean))
    assert per_site < 1e-10 and d_mean < 1e-10
    assert s.shape == (ns, Nx, Ny) and set(np.unique(s)) <= {0, 1}
    # E_loc = diagonal - Bx * (sum of N positive ratios): strictly below the diagonal energy for every sample
    sz = 2.0 * s - 1.0
    diag = -((sz[:, :-1, :] * sz[:, 1:, :]).sum(axis=(1, 2)) + (sz[:, :, :-1] * sz[:, :, 1:]).sum(axis=(1, 2)))
    assert np.all(e < diag)
    m = out["moments"]
    assert abs(m[0] / m[2] - e.mean()) < 1e-9 * abs(e.mean()) and m[2] == ns
    # the flipped-configuration queue of the reference (N+1 rows per sample) through log_prob on a few samples:
    # sum_i exp(lp(flip i) - lp) reproduces (diag - E_loc) / Bx
    k = 3
    lp = wf.log_prob(s[:k].astype(np.int32))
    acc = np.zeros(k)
    for i in range(Nx):
        for j in range(Ny):
            f = s[:k].astype(np.int32).copy()
            f[:, i, j] ^= 1
            acc += np.exp(0.5 * (wf.log_prob(f) - lp))
    assert np.allclose((diag[:k] - e[:k]) / 3.0, acc, rtol=1e-9)
