"""Per-tensor check of the complex RNN's gradient against float64 finite differences of the oracle's cost, at the reference run script's
size and on initial (unscaled) weights - every tensor relative to ITS OWN largest entry (the unit test normalises by the global maximum).
python tests/diagnostics/j1j2_gradient_per_tensor.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import models as M
from rnnwavefunctions_amd import _lib, params as P
from rnnwavefunctions_amd.training import cost_gradient
N, H, ns = 10, 10, 200
scope = "RNNwavefunction"
for seed, trained in ((111, 0), (111, 300)):
    prm = P.init_gru_params([H], seed=seed, scope=scope, heads=("wf_dense_ampl", "wf_dense_phase"))
    wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,))
    wf.set_params(prm, scope=scope)
    coup = np.concatenate([np.ones(N), 0.2 * np.ones(N), np.zeros(N), [0.0, 0.0]])
    if trained:
        wf.train_steps(ns, seed, 0, coup, [5e-4] * trained)
        prm = wf.get_params_dict(prm, scope)
    out = wf.vmc_step(ns, seed=3, step=9999, couplings=coup, want_samples=True, want_eloc=True)
    s, e = out["samples"], out["eloc"].astype(np.complex128)
    g = cost_gradient(wf, prm, scope, e.mean(), ns)
    prm64 = {k: v.astype(np.float64) for k, v in prm.items()}
    def cost():
        la = M.crnn_log_amplitude(prm64, s, dtype=np.float64)
        return 2 * np.real(np.mean(np.conj(la) * e) - np.conj(np.mean(la)) * np.mean(e))
    print("after %d training steps, <E> = %.4f" % (trained, e.mean().real))
    for name, gt in g.items():
        flat = prm64[name].ravel()
        fd = np.zeros(flat.size)
        for idx in range(flat.size):
            old = flat[idx]
            flat[idx] = old + 1e-5; cp = cost()
            flat[idx] = old - 1e-5; cm = cost()
            flat[idx] = old
            fd[idx] = (cp - cm) / 2e-5
        gg = gt.ravel()
        print("  %-62s max|g| %.3e  max|g-fd| %.3e  rel %.2e   corr %.6f" % (name[len(scope) + 1:], np.abs(gg).max(), np.abs(gg - fd).max(),
              np.abs(gg - fd).max() / max(np.abs(fd).max(), 1e-300), np.corrcoef(gg, fd)[0, 1] if gg.size > 2 and gg.std() > 0 else 1.0))
