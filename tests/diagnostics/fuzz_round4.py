"""Random configurations of the two round-4 paths (GPU box):
  (1) the bf16x3 layer pipeline for stacked layers (2..4 layers, UNEQUAL widths whose largest is 37..50, pRNN / parity / cRNN; the engine is
      forced so that small batches take it): local energies against the float64 oracle, gradients against its finite differences;
  (2) device-resident training (rnnwf_train_steps) against the host optimizer, every driver, random sizes / widths / layer counts: 12
      iterations, energies and final parameters must be identical.
python tests/diagnostics/fuzz_round4.py SEED TRIALS"""
import sys, os
os.environ["RNNWF_ENGINE"] = "bf16x3"
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import models as M
from oracle import estimators as E
from rnnwavefunctions_amd import _lib, params as P, training as T
from rnnwavefunctions_amd.training import cost_gradient
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SCOPE = "RNNwavefunction"
bad = 0
for trial in range(trials):
    model = rng.choice(["gru", "crnn", "parity"])
    L = int(rng.randint(2, 5))
    units = [int(rng.randint(3, 51)) for _ in range(L)]
    units[int(rng.randint(0, L))] = int(rng.randint(37, 51))
    units = tuple(units)
    N = int(rng.choice([6, 10, 14]))
    ns = int(rng.randint(5, 70))
    heads = ("wf_dense_ampl", "wf_dense_phase") if model == "crnn" else ("wf_dense",)
    prm = P.randomize_biases(P.scale_kernels(P.init_gru_params(list(units), seed=trial + 50, heads=heads), 1.4), trial)
    prm64 = {k: v.astype(np.float64) for k, v in prm.items()}
    mid = {"gru": _lib.MODEL_GRU1D, "crnn": _lib.MODEL_CRNN_U1, "parity": _lib.MODEL_GRU1D_PARITY}[model]
    wf = _lib.NativeWavefunction(mid, N, 1, units)
    wf.set_params(prm, scope=SCOPE)
    eng = wf.engine_name() if hasattr(wf, "engine_name") else "?"
    if model == "crnn":
        coup = np.concatenate([np.ones(N), 0.4 * np.ones(N), np.zeros(N), [0.0, 0.0]])
    else:
        coup = np.append(np.ones(N), 1.3)
    out = wf.vmc_step(ns, seed=trial, step=0, couplings=coup, want_samples=True, want_eloc=True)
    s = out["samples"].reshape(ns, N)
    if model == "crnn":
        e64 = E.j1j2_local_energies(np.ones(N), 0.4 * np.ones(N), np.zeros(N), s, lambda x: M.crnn_log_amplitude(prm64, x, dtype=np.float64), False, False)
        e = out["eloc"].astype(np.complex128)
        cost = lambda: 2 * np.real(np.mean(np.conj(M.crnn_log_amplitude(prm64, s, dtype=np.float64)) * e) - np.conj(np.mean(M.crnn_log_amplitude(prm64, s, dtype=np.float64))) * np.mean(e))
    else:
        lpf = (lambda x: M.prnn_paritysym_log_probability(prm64, x, dtype=np.float64)) if model == "parity" else (lambda x: M.prnn_log_probability(prm64, x, dtype=np.float64))
        e64 = E.ising_local_energies(np.ones(N), 1.3, s, lpf)
        e = out["eloc"]
        cost = lambda: np.mean(lpf(s) * e) - np.mean(e) * np.mean(lpf(s))
    err_e = np.abs(out["eloc"] - e64).max() / max(1.0, np.abs(e64).max())
    g = cost_gradient(wf, prm, SCOPE, e.mean(), ns)
    scale = max(np.abs(v).max() for v in g.values())
    worst = 0.0
    for name in g:
        flat = prm64[name].ravel()
        for idx in rng.choice(flat.size, size=min(flat.size, 2), replace=False):
            old = flat[idx]; eps = 1e-5
            flat[idx] = old + eps; cp = cost(); flat[idx] = old - eps; cm = cost(); flat[idx] = old
            worst = max(worst, abs((cp - cm) / (2 * eps) - g[name].ravel()[idx]) / scale)
    ok = err_e < 5e-5 and worst < 3e-3
    bad += not ok
    print("pipeline %-6s units=%-18s N=%2d ns=%2d engine=%s  E_loc err %.1e  grad-FD %.1e  %s" % (model, units, N, ns, eng, err_e, worst, "ok" if ok else "FAIL"), flush=True)

for trial in range(trials):
    kind = rng.choice(["tfim", "tfim_parity", "j1j2", "2d1d", "2d2d"])
    L = int(rng.randint(1, 4))
    H = int(rng.choice([5, 10, 16, 20, 33, 37, 44, 50, 52, 64, 68, 84, 100]))
    ns = int(rng.choice([50, 100, 200, 333]))
    seed = int(rng.randint(1, 1000))
    if kind == "2d2d":
        H = min(H, 84)
        kw = dict(systemsize_x=int(rng.randint(2, 5)), systemsize_y=int(rng.randint(2, 5)), num_units=H, numsamples=ns, learningrate=5e-3)
        run = T.run_2DTFIM_2DRNN
    elif kind == "2d1d":
        H = min(H, 68)
        kw = dict(systemsize_x=int(rng.randint(2, 5)), systemsize_y=int(rng.randint(2, 4)), num_units=H, num_layers=L, numsamples=ns, learningrate=1e-3)
        run = T.run_2DTFIM_1DRNN
    elif kind == "j1j2":
        kw = dict(systemsize=int(rng.choice([6, 8, 10, 12])), num_units=H, num_layers=L, numsamples=ns, learningrate=5e-4, J2_=float(rng.choice([0.0, 0.2, 0.5])),
                  Marshall_sign=bool(rng.randint(0, 2)))
        run = T.run_J1J2
    else:
        kw = dict(systemsize=int(rng.randint(5, 15)), num_units=H, num_layers=L, numsamples=ns, learningrate=5e-3, parity_symmetric=(kind == "tfim_parity"))
        run = T.run_1DTFIM
    res = {}
    try:
        for mode in (True, False):
            T.DEVICE_TRAINING = mode
            e, v = run(numsteps=12, seed=seed, verbose=False, **kw)
            res[mode] = (np.array(e), np.array(v), dict(run.last_params))
        ok = np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1]) and \
            all(np.array_equal(res[True][2][k], x) for k, x in res[False][2].items()) and res[True][0][0] != res[True][0][-1]
        note = "E %.6f -> %.6f" % (np.real(res[True][0][0]), np.real(res[True][0][-1]))
    except Exception as ex:          # a refusal (width / layer limits) is not a failure of the comparison; anything else is
        ok = "not implemented" in str(ex) or "too large" in str(ex) or "must be" in str(ex)
        note = "refused: " + str(ex)[:90]
    bad += not ok
    print("training %-11s %s  %s  %s" % (kind, kw, note, "ok" if ok else "FAIL"), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
