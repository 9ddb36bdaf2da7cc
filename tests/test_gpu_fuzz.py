"""Randomised size sweep on the GPU (was tools/fuzz_gpu.py): every width at a layout-class boundary of both engines, odd
batch sizes and chain lengths, pRNN / TFIM and cRNN / J1-J2, each against the oracle.

Bounds: log-probabilities |hip - f64 oracle| <= 3e-6 N + 3e-6; local energies |hip - oracle| <= 5e-5 |E| + 5e-6 N - relative
OR absolute, because an E_loc may sit near zero (its log-probabilities then still agree to 1e-6; a purely relative bound fails
there on both engines alike)."""
import numpy as np
import pytest

from oracle import estimators as E
from oracle import models as M
from rnnwavefunctions_amd import params as P

pytestmark = pytest.mark.gpu

WIDTHS = [1, 2, 3, 4, 5, 16, 17, 19, 20, 21, 35, 36, 37, 48, 49, 50, 51, 52, 53, 60, 67, 68, 69, 84, 99, 100]


def test_prnn_tfim_random_sizes_on_both_engines(monkeypatch):
    from rnnwavefunctions_amd import _lib
    rng = np.random.RandomState(123)
    worst = (0.0, None)
    for trial in range(2 * len(WIDTHS)):
        H = WIDTHS[trial % len(WIDTHS)]
        N = int(rng.choice([2, 3, 5, 8, 13, 21, 32, 33, 47]))
        ns = int(rng.choice([1, 7, 16, 17, 31, 32, 33, 63, 100, 257]))
        prm = P.randomize_biases(P.scale_kernels(P.init_gru_params([H], seed=trial), 1.5), trial + 1)
        prm64 = {k: v.astype(np.float64) for k, v in prm.items()}
        s = rng.randint(0, 2, (ns, N)).astype(np.int32)
        Jz = 1 + 0.1 * rng.standard_normal(N)
        e_ref, lp_ref = E.ising_local_energies(Jz, 0.8, s, lambda x: M.prnn_log_probability(prm64, x, dtype=np.float64),
                                               return_log_probs=True)
        for eng in ("auto", "bf16x3"):
            if eng == "bf16x3":
                monkeypatch.setenv("RNNWF_ENGINE", "bf16x3")
            else:
                monkeypatch.delenv("RNNWF_ENGINE", raising=False)
            wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, N, 1, (H,))       # the environment is read once, at rnnwf_create
            wf.set_params(prm, scope="RNNwavefunction")
            lp = np.zeros((N + 1) * ns)
            e = wf.tfim_eloc(s, Jz, 0.8, log_probs=lp)
            err = np.abs(lp - lp_ref.ravel()).max()
            de = (np.abs(e - e_ref) - 5e-5 * np.abs(e_ref)).max()
            tag = "trial %d %s H=%d N=%d ns=%d (%s)" % (trial, eng, H, N, ns, wf.engine_name())
            assert err <= 3e-6 * N + 3e-6, tag + ": max |lp - f64| = %.2e" % err
            assert de <= 5e-6 * N, tag + ": |E_loc - f64| beyond 5e-5 |E| + 5e-6 N by %.2e" % de
            if err > worst[0]:
                worst = (err, tag)
        monkeypatch.delenv("RNNWF_ENGINE", raising=False)
    print("pRNN sweep: worst |lp - f64| = %.2e at %s" % worst)


def test_crnn_j1j2_random_sizes_on_both_engines(monkeypatch):
    from rnnwavefunctions_amd import _lib
    rng = np.random.RandomState(321)
    heads = ("wf_dense_ampl", "wf_dense_phase")
    for trial, H in enumerate([2, 5, 19, 20, 21, 36, 37, 44, 49, 50, 51, 52, 53, 60, 68, 69, 100]):
        N = int(rng.choice([4, 6, 10, 16, 22]))
        ns = int(rng.choice([1, 9, 32, 33, 70]))
        prm = P.randomize_biases(P.scale_kernels(P.init_gru_params([H], seed=trial, heads=heads), 1.5), trial + 1)
        s = np.array([rng.permutation(np.r_[np.ones(N // 2), np.zeros(N // 2)]) for _ in range(ns)]).astype(np.int32)
        J1 = 1 + 0.1 * rng.standard_normal(N)
        J2 = 0.4 * np.ones(N)
        Bz = 0.05 * rng.standard_normal(N)
        e_ref = E.j1j2_local_energies(J1, J2, Bz, s, lambda x: M.crnn_log_amplitude(prm, x), False, False)
        for eng in ("f32", "bf16x3"):
            monkeypatch.setenv("RNNWF_ENGINE", eng)
            wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,))
            wf.set_params(prm, scope="RNNwavefunction")
            e, _ = wf.j1j2_eloc(s, J1, J2, Bz, False, False)
            assert np.allclose(e, e_ref, rtol=1e-4, atol=1e-4), "cRNN %s H=%d N=%d ns=%d: %.2e" % (eng, H, N, ns, np.abs(e - e_ref).max())
    monkeypatch.delenv("RNNWF_ENGINE", raising=False)
