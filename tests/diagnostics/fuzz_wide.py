"""Random widths, layer counts and models (one to four layers of any widths, single layers up to 150 units): local energies against the
float64 oracle, gradients against its finite differences.  python tests/diagnostics/fuzz_wide.py SEED TRIALS (GPU box)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import models as M
from oracle import estimators as E
from rnnwavefunctions_amd import _lib, params as P
from rnnwavefunctions_amd.training import cost_gradient
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
SCOPE = "RNNwavefunction"
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    model = rng.choice(["gru", "crnn", "parity", "gru64"])
    L = int(rng.randint(1, 5))
    if L == 1:
        units = (int(rng.choice([7, 33, 52, 53, 69, 100, 101, 120, 133, 150])) if model != "gru64" else int(rng.choice([9, 40, 68, 69, 90, 100])),)
    else:
        wmax = 68 if model == "gru64" else 100
        units = tuple(int(rng.randint(3, wmax + 1)) for _ in range(L))
    N = int(rng.choice([4, 6, 8])) if max(units) > 60 else int(rng.choice([6, 10, 14]))
    if model == "crnn" and N % 2: N += 1
    ns = int(rng.randint(5, 40))
    heads = ("wf_dense_ampl", "wf_dense_phase") if model == "crnn" else ("wf_dense",)
    dt = np.float64 if model == "gru64" else np.float32
    prm = P.randomize_biases(P.scale_kernels(P.init_gru_params(list(units), seed=trial + 5, heads=heads, dtype=dt), 1.4), trial)
    prm64 = {k: v.astype(np.float64) for k, v in prm.items()}
    mid = {"gru": _lib.MODEL_GRU1D, "crnn": _lib.MODEL_CRNN_U1, "parity": _lib.MODEL_GRU1D_PARITY, "gru64": _lib.MODEL_GRU1D_F64}[model]
    nx, ny = (N // 2, 2) if model == "gru64" else (N, 1)
    wf = _lib.NativeWavefunction(mid, nx, ny, units)
    wf.set_params(prm, scope=SCOPE)
    if model == "crnn":
        coup = np.concatenate([np.ones(N), 0.4 * np.ones(N), np.zeros(N), [0.0, 0.0]])
    else:
        coup = np.append(np.ones(N), 1.3)
    out = wf.vmc_step(ns, seed=trial, step=0, couplings=coup, want_samples=True, want_eloc=True)
    s = out["samples"].reshape(ns, N)
    if model == "crnn":
        la = M.crnn_log_amplitude(prm64, s, dtype=np.float64)
        e64 = E.j1j2_local_energies(np.ones(N), 0.4 * np.ones(N), np.zeros(N), s, lambda x: M.crnn_log_amplitude(prm64, x, dtype=np.float64), False, False)
        e = out["eloc"].astype(np.complex128)
        cost = lambda: 2 * np.real(np.mean(np.conj(M.crnn_log_amplitude(prm64, s, dtype=np.float64)) * e) - np.conj(np.mean(M.crnn_log_amplitude(prm64, s, dtype=np.float64))) * np.mean(e))
    else:
        lpf = (lambda x: M.prnn_paritysym_log_probability(prm64, x, dtype=np.float64)) if model == "parity" else (lambda x: M.prnn_log_probability(prm64, x, dtype=np.float64))
        if model == "gru64":      # the float64 GRU scores the open square lattice nx x ny (2DTFIM_1DRNN), raster order
            e64 = E.ising2d_local_energies(np.ones((nx, ny)), 1.3, nx, ny, s, lpf)
        else:
            e64 = E.ising_local_energies(np.ones(N), 1.3, s, lpf)
        e = out["eloc"]
        cost = lambda: np.mean(lpf(s) * e) - np.mean(e) * np.mean(lpf(s))
    err_e = np.abs(out["eloc"] - e64).max() / max(1.0, np.abs(e64).max())
    g = cost_gradient(wf, prm, SCOPE, e.mean(), ns)
    scale = max(np.abs(v).max() for v in g.values())
    worst = 0.0
    for name in g:
        flat = prm64[name].ravel()
        for idx in rng.choice(flat.size, size=min(flat.size, 3), replace=False):
            old = flat[idx]; eps = 1e-5 if model != "gru64" else 1e-6
            flat[idx] = old + eps; cp = cost(); flat[idx] = old - eps; cm = cost(); flat[idx] = old
            worst = max(worst, abs((cp - cm) / (2 * eps) - g[name].ravel()[idx]) / scale)
    ok = err_e < (1e-9 if model == "gru64" else 5e-5) and worst < (1e-6 if model == "gru64" else 3e-3)
    bad += not ok
    print("%-6s units=%-22s N=%2d ns=%2d  E_loc err %.1e  grad-FD %.1e  %s" % (model, units, N, ns, err_e, worst, "ok" if ok else "FAIL"), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
