"""Where a training iteration at the reference run script's size (N = 20, 50 units, 500 samples) spends its time: python tools/iter_breakdown.py"""
import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from rnnwavefunctions_amd import _lib, params as P
from rnnwavefunctions_amd.training import cost_gradient, Adam
N, H, ns = 20, 50, 500
scope = "RNNwavefunction"
prm = P.init_gru_params([H], seed=111, scope=scope)
wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, N, 1, (H,))
wf.set_params(prm, scope=scope)
opt = Adam()
coup = np.append(np.ones(N), 1.0)
T = dict(step=0.0, grad=0.0, adam=0.0, setp=0.0)
for it in range(260):
    t0 = time.perf_counter()
    m = wf.vmc_step(ns, seed=111, step=it, couplings=coup)["moments"]
    t1 = time.perf_counter()
    g = cost_gradient(wf, prm, scope, m[0] / m[2], m[2])
    t2 = time.perf_counter()
    prm = opt.step(prm, g, 5e-3)
    t3 = time.perf_counter()
    wf.set_params(prm, scope=scope)
    t4 = time.perf_counter()
    if it >= 60:
        T["step"] += t1 - t0; T["grad"] += t2 - t1; T["adam"] += t3 - t2; T["setp"] += t4 - t3
print({k: round(v / 200 * 1e3, 4) for k, v in T.items()}, "ms per iteration; sum", round(sum(T.values()) / 200 * 1e3, 4))
