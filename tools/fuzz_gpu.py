"""Randomised size sweep on the GPU box (not part of the test suite): widths at every layout-class boundary, odd batch
sizes, both flip engines, pRNN/TFIM and cRNN/J1-J2, each against the float64 / f32 oracle.
   python tools/fuzz_gpu.py
(An E_loc whose value happens to sit near zero can exceed the RELATIVE bound while its log-probabilities agree to 1e-6:
trial 43 does, on both engines alike.)"""
import sys, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from oracle import models as M, estimators as E
from rnnwavefunctions_amd import _lib, params as P
rng = np.random.RandomState(123)
bad = 0
Hs = [1, 2, 3, 4, 5, 16, 17, 19, 20, 21, 35, 36, 37, 48, 49, 50, 51, 52, 53, 67, 68, 69, 84, 99, 100]
for trial in range(60):
    H = Hs[trial % len(Hs)]
    N = int(rng.choice([2, 3, 5, 8, 13, 21, 32, 33, 47]))
    ns = int(rng.choice([1, 7, 16, 17, 31, 32, 33, 63, 100, 257]))
    prm = P.randomize_biases(P.scale_kernels(P.init_gru_params([H], seed=trial), 1.5), trial + 1)
    prm64 = {k: v.astype(np.float64) for k, v in prm.items()}
    wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, N, 1, (H,))
    wf.set_params(prm, scope="RNNwavefunction")
    s = rng.randint(0, 2, (ns, N)).astype(np.int32)
    Jz = 1 + 0.1 * rng.standard_normal(N)
    for eng in ("auto", "bf16x3"):
        import os
        if eng == "bf16x3":
            os.environ["RNNWF_ENGINE"] = "bf16x3"; wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, N, 1, (H,)); wf.set_params(prm, scope="RNNwavefunction")
        else:
            os.environ.pop("RNNWF_ENGINE", None)
        lp = np.zeros((N + 1) * ns)
        e = wf.tfim_eloc(s, Jz, 0.8, log_probs=lp)
        e_ref, lp_ref = E.ising_local_energies(Jz, 0.8, s, lambda x: M.prnn_log_probability(prm64, x, dtype=np.float64), return_log_probs=True)
        err = np.abs(lp - lp_ref.ravel()).max(); rel = np.abs(e / e_ref - 1).max()
        ok = err <= 3e-6 * N + 3e-6 and rel < 5e-5
        if not ok:
            bad += 1
            print("FAIL", trial, eng, "H", H, "N", N, "ns", ns, "err", err, "rel", rel, wf.engine_name())
    os.environ.pop("RNNWF_ENGINE", None)
print("fuzz done, failures:", bad)

# ---- complex RNN / J1-J2: widths at the layout-class boundaries, both engines ------------------------------------
bad2 = 0
for trial, H in enumerate([2, 5, 19, 20, 21, 36, 37, 44, 49, 50, 51, 52, 53, 68, 69, 100]):
    N = int(rng.choice([4, 6, 10, 16, 22]))
    ns = int(rng.choice([1, 9, 32, 33, 70]))
    heads = ("wf_dense_ampl", "wf_dense_phase")
    prm = P.randomize_biases(P.scale_kernels(P.init_gru_params([H], seed=trial, heads=heads), 1.5), trial + 1)
    s = np.array([rng.permutation(np.r_[np.ones(N // 2), np.zeros(N // 2)]) for _ in range(ns)]).astype(np.int32)
    J1 = 1 + 0.1 * rng.standard_normal(N); J2 = 0.4 * np.ones(N); Bz = 0.05 * rng.standard_normal(N)
    e_ref = E.j1j2_local_energies(J1, J2, Bz, s, lambda x: M.crnn_log_amplitude(prm, x), False, False)
    for eng in ("f32", "bf16x3"):
        os.environ["RNNWF_ENGINE"] = eng
        wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,))
        wf.set_params(prm, scope="RNNwavefunction")
        e, _ = wf.j1j2_eloc(s, J1, J2, Bz, False, False)
        if not np.allclose(e, e_ref, rtol=1e-4, atol=1e-4):
            bad2 += 1
            print("FAIL cRNN", eng, "H", H, "N", N, "ns", ns, np.abs(e - e_ref).max())
    os.environ.pop("RNNWF_ENGINE", None)
print("cRNN fuzz done, failures:", bad2)
