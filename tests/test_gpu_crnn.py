"""GPU parity tests of the complex-RNN / J1-J2 path (HIP through the C ABI) vs the CPU oracle.

Tolerances (the reference computes this path in float32 / complex64):
  log psi     : |hip - oracle| <= 3e-6 * N + 3e-6 on both parts (phases compared modulo nothing: both are sums
                of per-site principal values)
  E_loc       : |hip - oracle| <= 5e-5 * (1 + |E|) per sample (complex64 output)
"""

import numpy as np
import pytest

from conftest import golden_params
from oracle import estimators as E
from oracle import models as M
from oracle import philox
from rnnwavefunctions_amd import params as P

pytestmark = pytest.mark.gpu
HEADS = ("wf_dense_ampl", "wf_dense_phase")


def make_wf(N, H, prm):
    from rnnwavefunctions_amd import _lib
    wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,))
    wf.set_params(prm, scope="RNNwavefunction")
    return wf


def trained_like(H, seed):
    return P.randomize_biases(P.scale_kernels(P.init_gru_params([H], seed=seed, heads=HEADS), 2.0), seed + 1)


def zero_mag_batch(ns, N, seed):
    rng = np.random.RandomState(seed)
    return np.stack([rng.permutation(np.repeat([0, 1], N // 2)) for _ in range(ns)]).astype(np.int32)


@pytest.mark.parametrize("N,H,B", [(10, 11, 50), (40, 50, 100), (34, 20, 70), (16, 100, 33), (12, 128, 24), (8, 200, 20), (8, 256, 17)])
def test_log_amplitude_matches_oracle(N, H, B):
    prm = trained_like(H, seed=H)
    wf = make_wf(N, H, prm)
    s = zero_mag_batch(B, N, 3)
    got = wf.log_amp(s)
    ref = M.crnn_log_amplitude(prm, s)
    assert got.dtype == np.complex64
    err_re, err_im = np.abs(got.real - ref.real).max(), np.abs(got.imag - ref.imag).max()
    print("N=%d H=%d: |d re|=%.2e |d im|=%.2e" % (N, H, err_re, err_im))
    tol = 3e-6 * N + 3e-6
    assert err_re <= tol and err_im <= 4 * tol
    assert np.allclose(wf.log_prob(s), 2.0 * ref.real.astype(np.float64), atol=2 * tol)
    # outside the zero-magnetisation sector the amplitude vanishes: Re log psi = -inf
    bad = s.copy()
    bad[:, 0] = bad[:, 1] = bad[:, 2] = 1
    bad = bad[bad.sum(axis=1) != N // 2]
    assert np.all(np.isneginf(wf.log_amp(bad).real))


def test_masked_sampler_matches_oracle_and_conserves_magnetisation():
    N, H, ns = 40, 50, 1000
    prm = trained_like(H, seed=7)
    wf = make_wf(N, H, prm)
    s, lg = wf.sample(ns, seed=5, step=2, return_log=True)
    assert np.all(s.sum(axis=1) == N // 2)
    s_ref = M.crnn_sample(prm, N, philox.uniforms(5, 2, 0, ns, N))
    bad = (s != s_ref).any(axis=1).sum()
    print("cRNN sampler: %d of %d rows differ from the oracle" % (bad, ns))
    assert bad <= 2
    ok = ~(s != s_ref).any(axis=1)
    assert np.allclose(lg[ok], 2 * M.crnn_log_amplitude(prm, s_ref[ok]).real, atol=3e-4)
    assert np.array_equal(np.concatenate([wf.sample(300, 5, 2, 0), wf.sample(700, 5, 2, 300)]), s)


def test_j1j2_eloc_matches_reference_golden(golden_estimators):
    """G4d: reference J1J2Slices + the reference's E_loc expression on oracle log-amplitudes."""
    g = golden_estimators
    prm = golden_params(g, "g4d")
    s = g["g4d_samples"]
    N = s.shape[1]
    wf = make_wf(N, 11, prm)
    e, ncon = wf.j1j2_eloc(s, np.ones(N), float(g["g4d_J2"]) * np.ones(N), np.zeros(N))
    assert ncon == int(g["g4d_offsets"][-1])
    print("G4d: max |E diff| = %.2e" % np.abs(e - g["g4d_eloc"]).max())
    assert np.allclose(e, g["g4d_eloc"], rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("periodic,marshall,J2v", [(False, False, 0.5), (True, False, 0.5), (False, True, 0.2),
                                                   (True, True, 0.8), (False, False, 0.0)])
def test_j1j2_eloc_flags(periodic, marshall, J2v):
    N, H, ns = 16, 20, 77
    prm = trained_like(H, seed=3)
    wf = make_wf(N, H, prm)
    s = zero_mag_batch(ns, N, 11)
    rng = np.random.RandomState(2)
    J1 = 1.0 + 0.1 * rng.standard_normal(N)
    J2 = J2v * np.ones(N)
    Bz = 0.05 * rng.standard_normal(N)
    e, ncon = wf.j1j2_eloc(s, J1, J2, Bz, periodic, marshall)
    e_ref = E.j1j2_local_energies(J1, J2, Bz, s, lambda x: M.crnn_log_amplitude(prm, x), periodic, marshall)
    _, _, offs = E.j1j2_slices(J1, J2, Bz, s, periodic, marshall)
    assert ncon == offs[-1]
    assert np.allclose(e, e_ref, rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("N,H,ns", [(16, 20, 60), (14, 50, 48), (12, 64, 40), (12, 100, 40), (10, 80, 33),
                                     # both sides of the bf16x3 engine's width-class boundaries (36|37, 50|51, 52|53, 68|69)
                                     (12, 36, 33), (14, 37, 40), (10, 51, 33), (12, 52, 33), (12, 53, 40), (14, 60, 33),
                                     (10, 68, 37), (12, 69, 33)])
def test_both_swap_engines_agree_with_the_f64_oracle(N, H, ns, monkeypatch):
    """The J1-J2 swap pass on the bf16x3 engine (default; above 68 units with the w3 fragments read through L2) and on the
    f32-input MFMA (RNNWF_ENGINE=f32): both against the float64 oracle at the f32 tolerance, and each other."""
    prm = trained_like(H, seed=N + H)
    prm64 = {k: v.astype(np.float64) for k, v in prm.items()}
    s = zero_mag_batch(ns, N, 5)
    rng = np.random.RandomState(4)
    J1 = 1.0 + 0.1 * rng.standard_normal(N)
    J2 = 0.4 * np.ones(N)
    Bz = np.zeros(N)
    e64 = E.j1j2_local_energies(J1, J2, Bz, s, lambda x: M.crnn_log_amplitude(prm64, x, dtype=np.float64), False, False)
    got = {}
    for engine in ("f32", "bf16x3"):
        monkeypatch.setenv("RNNWF_ENGINE", engine)
        wf = make_wf(N, H, prm)
        e, _ = wf.j1j2_eloc(s, J1, J2, Bz, False, False)
        assert wf.engine_name() == ("bf16x3" if engine == "bf16x3" else "f32mfma")
        got[engine] = e
        err = np.abs(e - e64).max() / max(1.0, np.abs(e64).max())
        print("cRNN N=%d H=%d %-6s: max |E_loc - f64| / max|E| = %.2e" % (N, H, engine, err))
        assert err < 3e-5
    assert np.allclose(got["f32"], got["bf16x3"], rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("N,H,ns", [(12, 128, 33), (8, 196, 20), (8, 256, 17)])
def test_local_energies_above_100_units(N, H, ns):
    """Above 100 units the weight image no longer fits LDS: it stays in global memory and the kernels (f32-input MFMA) read its
    fragments through L2 (GruLayout::SPILL).  Sampling respects the U(1) mask, local energies match the float64 oracle."""
    prm = trained_like(H, seed=N + H)
    prm64 = {k: v.astype(np.float64) for k, v in prm.items()}
    wf = make_wf(N, H, prm)
    rng = np.random.RandomState(4)
    J1, J2, Bz = 1.0 + 0.1 * rng.standard_normal(N), 0.4 * np.ones(N), np.zeros(N)
    s = zero_mag_batch(ns, N, 5)
    e, _ = wf.j1j2_eloc(s, J1, J2, Bz, False, False)
    assert wf.engine_name() == "f32mfma"
    e64 = E.j1j2_local_energies(J1, J2, Bz, s, lambda x: M.crnn_log_amplitude(prm64, x, dtype=np.float64), False, False)
    err = np.abs(e - e64).max() / max(1.0, np.abs(e64).max())
    print("cRNN N=%d H=%d: max |E_loc - f64| / max|E| = %.2e" % (N, H, err))
    assert err < 3e-5
    drawn = wf.sample(64, seed=3, step=0)
    assert np.all(drawn.sum(axis=1) == N // 2)


@pytest.mark.parametrize("H", [20, 36, 37, 50, 52, 53, 60, 64, 68, 69, 100])
def test_copies_of_one_configuration_get_identical_values(H, monkeypatch):
    """The complex RNN's swap pass: 80 copies of one zero-magnetisation configuration -> bit-identical local energies on
    either engine, launch after launch (any difference is a scheduling hazard)."""
    N = 16
    prm = trained_like(H, seed=H)
    s = np.repeat(zero_mag_batch(1, N, H), 80, axis=0)
    J1, J2, Bz = np.ones(N), 0.3 * np.ones(N), np.zeros(N)
    for engine in ("bf16x3", "f32"):
        monkeypatch.setenv("RNNWF_ENGINE", engine)
        wf = make_wf(N, H, prm)
        e0, _ = wf.j1j2_eloc(s, J1, J2, Bz, False, False)
        assert np.unique(e0).size == 1, (engine, np.unique(e0).size)
        for _ in range(3):
            assert np.array_equal(wf.j1j2_eloc(s, J1, J2, Bz, False, False)[0], e0)


def test_j1j2_slices_formulation_through_the_facade_equals_the_fused_call():
    """The reference's formulation - J1J2Slices, chunked log-amplitudes through `sess.run`, E_loc = sum_s H_s exp(log psi_s - log psi_0)
    (J1J2/TrainingRNN_J1J2.py:247-279; assembled by tests/graph_mode.py: slices_local_energies) - against the fused estimator."""
    from graph_mode import slices_local_energies
    from rnnwavefunctions_amd import compat as tf
    from rnnwavefunctions_amd.J1J2.TrainingRNN_J1J2 import J1J2_local_energies, J1J2Slices, RNNwavefunction
    N, batch = 12, 64
    J1, J2, Bz = np.ones(N), 0.5 * np.ones(N), np.zeros(N)
    wf = RNNwavefunction(N, units=[20], cell=tf.CudnnCompatibleGRUCell, seed=111)
    assert wf.num_params() == 3 * 400 + 3 * 2 * 20 + 4 * 20 + 2 * (2 * 20 + 2)     # 3h^2 + 3dh + 4h + two heads
    sess = tf.Session(graph=wf.graph)
    any_in = tf.placeholder(dtype=tf.int32, shape=(None, N))
    score = wf.log_amplitude(any_in, inputdim=2)
    drawn = sess.run(wf.sample(numsamples=batch, inputdim=2))
    assert np.all(drawn.sum(axis=1) == N // 2)
    e_slices, total = slices_local_energies(J1J2Slices, lambda rows: sess.run(score, feed_dict={any_in: rows}), J1, J2, Bz, drawn, chunk=500)
    fused, ncon = J1J2_local_energies(J1, J2, Bz, drawn, score, return_num_connected=True)
    assert ncon == total and np.allclose(fused, e_slices, rtol=5e-5, atol=5e-5)
    assert np.isfinite(np.mean(e_slices)) and np.var(np.real(e_slices)) >= 0


def test_vmc_step_j1j2():
    N, H, ns = 20, 50, 257
    prm = trained_like(H, seed=4)
    wf = make_wf(N, H, prm)
    couplings = np.concatenate([np.ones(N), 0.5 * np.ones(N), np.zeros(N), [0.0, 0.0]])
    out = wf.vmc_step(ns, seed=9, step=1, couplings=couplings, want_samples=True, want_eloc=True)
    s = out["samples"]
    assert np.array_equal(s, wf.sample(ns, seed=9, step=1))
    e, _ = wf.j1j2_eloc(s, np.ones(N), 0.5 * np.ones(N), np.zeros(N))
    assert np.allclose(out["eloc"], e, rtol=1e-6, atol=1e-6)
    m = out["moments"]
    assert m[2] == ns
    assert np.isclose(m[0] / ns, e.real.astype(np.float64).mean(), rtol=1e-9)
    assert np.isclose(m[3] / ns, e.imag.astype(np.float64).mean(), rtol=1e-6, atol=1e-9)
    assert np.isclose(m[1] / ns - (m[0] / ns) ** 2, np.var(e.real.astype(np.float64)), rtol=1e-6)


def test_config3_properties():
    """BASELINE config 3 (N=40, J2=0.5, num_units=50, numsamples=10000): zero magnetisation everywhere,
    finite local energies, connected-configuration count in the range the survey measured (~40 per sample),
    and E_loc / <E>/N against the oracle on 4 096 samples of the same sample matrix (~165 000 connected configurations
    through the NumPy oracle, in chunks)."""
    N, H, ns = 40, 50, 10000
    prm = P.init_gru_params([H], seed=111, heads=HEADS)
    wf = make_wf(N, H, prm)
    couplings = np.concatenate([np.ones(N), 0.5 * np.ones(N), np.zeros(N), [0.0, 0.0]])
    out = wf.vmc_step(ns, seed=111, step=0, couplings=couplings, want_samples=True, want_eloc=True)
    s, e = out["samples"], out["eloc"]
    assert np.all(s.sum(axis=1) == N // 2)
    assert np.all(np.isfinite(e.real)) and np.all(np.isfinite(e.imag))
    sub = np.arange(0, ns, 2)[:4096]
    e_ref = np.concatenate([E.j1j2_local_energies(np.ones(N), 0.5 * np.ones(N), np.zeros(N), s[sub[k:k + 512]],
                                                  lambda x: M.crnn_log_amplitude(prm, x)) for k in range(0, 4096, 512)])
    per_site = np.abs(e[sub] - e_ref).max() / N
    d_mean = abs(e[sub].astype(np.complex128).mean() - e_ref.astype(np.complex128).mean()) / N
    print("cfg3: over 4096 samples max |E_loc diff| / N = %.2e, |<E> diff| / N = %.2e" % (per_site, d_mean))
    assert per_site < 1e-5 and d_mean < 1e-5
    _, ncon = wf.j1j2_eloc(s, np.ones(N), 0.5 * np.ones(N), np.zeros(N))
    assert 30 * ns < ncon < 50 * ns


@pytest.mark.parametrize("N,H,ns", [(10, 11, 50), (40, 50, 333), (34, 20, 70), (16, 64, 33)])
def test_cooperative_base_pass_is_bit_identical(N, H, ns, monkeypatch):
    """The two f32-input-MFMA base kernels, crnn_base_coop_kernel vs crnn_base_kernel: samples, log-amplitudes and J1-J2 local
    energies bit for bit (RNNWF_BASE=f32: up to 52 units the default is the bf16 cooperative kernel, next test)."""
    prm = trained_like(H, seed=N)
    monkeypatch.setenv("RNNWF_BASE", "f32")
    wf = make_wf(N, H, prm)
    J1, J2, Bz = np.ones(N), 0.5 * np.ones(N), np.zeros(N)
    s1 = wf.sample(ns, seed=4, step=1)
    a1 = wf.log_amp(s1)
    e1, n1 = wf.j1j2_eloc(s1, J1, J2, Bz)
    monkeypatch.setenv("RNNWF_NO_COOP", "1")                 # read once, at rnnwf_create
    wf = make_wf(N, H, prm)
    monkeypatch.delenv("RNNWF_NO_COOP")
    monkeypatch.delenv("RNNWF_BASE")
    s2 = wf.sample(ns, seed=4, step=1)
    a2 = wf.log_amp(s1)
    e2, n2 = wf.j1j2_eloc(s1, J1, J2, Bz)
    assert np.array_equal(s1, s2) and np.array_equal(a1, a2)
    assert n1 == n2 and np.array_equal(e1, e2)
    assert np.all(s1.sum(axis=1) == N // 2)


@pytest.mark.parametrize("N,H,ns", [(10, 11, 50), (40, 50, 333), (34, 20, 700), (16, 36, 33), (22, 52, 5000), (40, 50, 30000)])
def test_bf16_cooperative_base_pass_against_the_f32_kernels_and_the_oracle(N, H, ns, monkeypatch):
    """The complex RNN's base pass on the bf16 matrix core (gru_kernels.h: coop_base_pass_bf, the default up to 52 units, every
    batch size): U(1) on every sample, log-amplitudes against the float64 oracle at the f32 tolerance and against the
    f32-input-MFMA kernels, the same draws except near-ties, shard invariance, the same J1-J2 local energies."""
    prm = trained_like(H, seed=N + 3)
    prm64 = {k: v.astype(np.float64) for k, v in prm.items()}
    wf = make_wf(N, H, prm)
    s1 = wf.sample(ns, seed=4, step=1)
    assert np.all(s1.sum(axis=1) == N // 2)
    cut = (ns // 3) | 1
    assert np.array_equal(np.concatenate([wf.sample(cut, seed=4, step=1), wf.sample(ns - cut, seed=4, step=1, sample_offset=cut)]), s1)
    sub = np.arange(0, ns, max(1, ns // 300))[:300]
    a1 = wf.log_amp(s1[sub])
    ref = M.crnn_log_amplitude(prm64, s1[sub], dtype=np.float64)
    assert np.abs(a1 - ref).max() <= 2e-6 * N + 2e-6
    monkeypatch.setenv("RNNWF_BASE", "f32")
    wf32 = make_wf(N, H, prm)
    monkeypatch.delenv("RNNWF_BASE")
    s2 = wf32.sample(ns, seed=4, step=1)
    bad = (s1 != s2).any(axis=1).sum()
    print("cRNN N=%d H=%d ns=%d: %d rows drawn differently by the bf16 and the f32 base pass" % (N, H, ns, bad))
    assert bad <= max(2, ns // 2000)
    a2 = wf32.log_amp(s1[sub])
    assert np.abs(a1 - a2).max() <= 2e-6 * N + 2e-6
    J1, J2, Bz = np.ones(N), 0.5 * np.ones(N), np.zeros(N)
    e1, n1 = wf.j1j2_eloc(s1[sub], J1, J2, Bz)
    e2, n2 = wf32.j1j2_eloc(s1[sub], J1, J2, Bz)
    assert n1 == n2 and np.allclose(e1, e2, rtol=5e-5, atol=5e-5)


# ---- stacked layers: the reference's DEFAULT constructor is units=[10, 10] (J1J2/ComplexRNNwavefunction.py:16,40) ----

def stacked_like(H, L, seed):
    return P.randomize_biases(P.scale_kernels(P.init_gru_params([H] * L, seed=seed, heads=HEADS), 1.6), seed + 1)


@pytest.mark.parametrize("N,H,L,B", [(12, 10, 2, 40), (10, 20, 3, 33), (20, 50, 2, 24), (8, 36, 3, 17), (16, 52, 2, 16),
                                      (12, 50, 3, 24),
                                      (10, 53, 2, 20), (8, 64, 3, 17), (10, 100, 2, 24), (8, 100, 3, 18),   # 53..100 units: upper images through L2
                                      (10, 20, 4, 24), (8, 64, 4, 17)])      # four layers
def test_stacked_layers_log_amplitude_sampling_and_eloc_match_oracle(N, H, L, B):
    from rnnwavefunctions_amd import _lib
    prm = stacked_like(H, L, seed=N + H)
    wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,) * L)
    wf.set_params(prm, scope="RNNwavefunction")
    assert wf.num_params() == P.count_params(prm)
    s = zero_mag_batch(B, N, 5)
    got, ref = wf.log_amp(s), M.crnn_log_amplitude(prm, s)
    tol = 3e-6 * N * L + 3e-6
    assert np.abs(got.real - ref.real).max() <= tol and np.abs(got.imag - ref.imag).max() <= 4 * tol
    smp = wf.sample(200, seed=3, step=1)
    assert np.all(smp.sum(axis=1) == N // 2)
    s_ref = M.crnn_sample(prm, N, philox.uniforms(3, 1, 0, 200, N))
    assert (smp != s_ref).any(axis=1).sum() <= 2
    J1, J2, Bz = np.ones(N), 0.5 * np.ones(N), np.zeros(N)
    e, ncon = wf.j1j2_eloc(s, J1, J2, Bz)
    e_ref = E.j1j2_local_energies(J1, J2, Bz, s, lambda x: M.crnn_log_amplitude(prm, x), False, False)
    assert np.allclose(e, e_ref, rtol=1e-4, atol=1e-4)
    out = wf.vmc_step(64, seed=9, step=0, couplings=np.concatenate([J1, J2, Bz, [0.0, 0.0]]), want_samples=True, want_eloc=True)
    e2, _ = wf.j1j2_eloc(out["samples"], J1, J2, Bz)
    assert np.allclose(out["eloc"], e2, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("N,H,L,B", [(20, 50, 2, 40), (12, 37, 2, 33), (10, 44, 3, 24), (8, 50, 4, 20), (34, 49, 2, 50), (2, 50, 2, 4)])
def test_stacked_layers_on_both_engines(N, H, L, B, monkeypatch):
    """The complex RNN's stack (default: two layers, J1J2/ComplexRNNwavefunction.py:16,40) at 37..50 units on the bf16x3 engine - a
    pipeline of one ping-pong kernel per layer (csrc/crnn_split_kernels.h: crnn_swap_pp_kernel<.., STACK> -> crnn_swap_pp_upper_kernel)
    - and on the f32-input MFMA: J1-J2 local energies of both against the oracle, all four flag combinations on the first case."""
    from rnnwavefunctions_amd import _lib
    prm = stacked_like(H, L, seed=N + H)
    s = zero_mag_batch(B, N, 6)
    rng = np.random.RandomState(N)
    J1, J2, Bz = 1.0 + 0.1 * rng.standard_normal(N), 0.5 + 0.1 * rng.standard_normal(N), 0.1 * rng.standard_normal(N)
    flags = [(False, False), (True, False), (False, True), (True, True)] if (N, H) == (20, 50) else [(False, False)]
    for periodic, marshall in flags:
        e_ref = E.j1j2_local_energies(J1, J2, Bz, s, lambda x: M.crnn_log_amplitude(prm, x), periodic, marshall)
        got = {}
        for engine in ("bf16x3", "f32"):
            monkeypatch.setenv("RNNWF_ENGINE", engine)
            wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,) * L)
            wf.set_params(prm, scope="RNNwavefunction")
            e, ncon = wf.j1j2_eloc(s, J1, J2, Bz, periodic=periodic, marshall=marshall)
            assert wf.engine_name() == ("bf16x3" if engine == "bf16x3" else "f32mfma")
            print("L=%d N=%d H=%d %s periodic=%d marshall=%d: max |dE| = %.2e" % (L, N, H, engine, periodic, marshall, np.abs(e - e_ref).max()))
            assert np.allclose(e, e_ref, rtol=1e-4, atol=1e-4)
            got[engine] = (e, ncon)
        assert got["bf16x3"][1] == got["f32"][1]
        assert np.allclose(got["bf16x3"][0], got["f32"][0], rtol=5e-5, atol=5e-5)


def test_stacked_layers_bf16x3_vmc_step_at_speed_size():
    """The layer pipeline at a batch it is chosen for by default: N=24, units=[50,50], 4 096 samples; 256 of them against the oracle,
    zero magnetisation, shard invariance bit for bit."""
    from rnnwavefunctions_amd import _lib
    N, H, L, ns = 24, 50, 2, 4096
    prm = stacked_like(H, L, seed=3)
    wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,) * L)
    wf.set_params(prm, scope="RNNwavefunction")
    J1, J2, Bz = np.ones(N), 0.5 * np.ones(N), np.zeros(N)
    c = np.concatenate([J1, J2, Bz, [0.0, 0.0]])
    out = wf.vmc_step(ns, seed=5, step=1, couplings=c, want_samples=True, want_eloc=True)
    assert wf.engine_name() == "bf16x3"
    s, e = out["samples"], out["eloc"]
    assert np.all(s.sum(axis=1) == N // 2)
    sub = np.arange(0, ns, 16)
    e_ref = E.j1j2_local_energies(J1, J2, Bz, s[sub], lambda x: M.crnn_log_amplitude(prm, x), False, False)
    assert np.allclose(e[sub], e_ref, rtol=1e-4, atol=1e-4)
    lo = wf.vmc_step(ns // 2, seed=5, step=1, couplings=c, want_eloc=True)
    hi = wf.vmc_step(ns // 2, seed=5, step=1, couplings=c, sample_offset=ns // 2, want_eloc=True)
    assert np.array_equal(np.concatenate([lo["eloc"], hi["eloc"]]), e)


def test_default_constructor_of_the_complex_wave_function():
    """RNNwavefunction(systemsize, cell) with every other argument at the reference's default: units=[10, 10]."""
    from rnnwavefunctions_amd import compat as tf
    from rnnwavefunctions_amd.J1J2.ComplexRNNwavefunction import RNNwavefunction
    N = 12
    wf = RNNwavefunction(N, cell=tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell)
    assert wf.units == [10, 10]
    # two GRU layers (3h(d+h)+... as the TF variables count them) + two Dense(2) heads
    h = 10
    assert wf.num_params() == (2 + h) * 2 * h + 2 * h + 2 * h + h + h * h + h + (h + h) * 2 * h + 2 * h + h * h + h + h * h + h + 2 * (2 * h + 2)
    sess = tf.Session(graph=wf.graph)
    samples = sess.run(wf.sample(numsamples=50, inputdim=2))
    assert samples.shape == (50, N) and np.all(samples.sum(axis=1) == N // 2)
    ph = tf.placeholder(tf.int32, shape=(None, N))
    la = sess.run(wf.log_amplitude(ph, inputdim=2), feed_dict={ph: samples})
    assert np.allclose(la, M.crnn_log_amplitude(wf.get_params(), samples.astype(np.int32)), atol=1e-4)
    assert len(wf.rnn.variables) == 12 and wf.dense_ampl.count_params() == 22 and wf.dense_phase.count_params() == 22
    # and it trains: the gradient of the two-layer stack (tests/test_gpu_training.py checks it against finite differences)
    wf._native.vmc_step(16, seed=1, step=0, couplings=np.concatenate([np.ones(N), np.zeros(2 * N), [0.0, 0.0]]))
    g = wf._native.vmc_gradient(0.0, 16, {"wf_dense_ampl/kernel": (10, 2)})
    assert np.isfinite(g["wf_dense_ampl/kernel"]).all() and np.abs(g["wf_dense_ampl/kernel"]).max() > 0
