import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import models as M, estimators as E
from rnnwavefunctions_amd import _lib, params as P
N, H, ns = 1000, 50, 512
prm = P.init_gru_params([H], seed=111)
wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, N, 1, (H,))
wf.set_params(prm, scope="RNNwavefunction")
t0 = time.perf_counter()
out = wf.vmc_step(ns, seed=1, step=0, couplings=np.append(np.ones(N), 1.0), want_samples=True, want_eloc=True)
t1 = time.perf_counter()
out = wf.vmc_step(ns, seed=1, step=1, couplings=np.append(np.ones(N), 1.0), want_samples=True, want_eloc=True)
t2 = time.perf_counter()
s, e = out["samples"], out["eloc"]
sub = [0, 255, 511]
e_ref = E.ising_local_energies(np.ones(N), 1.0, s[sub], lambda x: M.prnn_log_probability(prm, x))
print("N=1000: step %.1f ms (first %.1f), <E>/N = %.5f (DMRG -1.27288), max|dE|/N = %.2e, engine %s" % ((t2 - t1) * 1e3, (t1 - t0) * 1e3, e.mean() / N, np.abs(e[sub] - e_ref).max() / N, wf.engine_name()))
