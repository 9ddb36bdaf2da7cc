"""The same breakdown at config 3's and config 2's sizes: python tools/iter_breakdown_big.py"""
import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from rnnwavefunctions_amd import _lib, params as P
from rnnwavefunctions_amd.training import cost_gradient, Adam
def run(model, N, H, ns, heads, coup, cplx):
    scope = "RNNwavefunction"
    prm = P.init_gru_params([H], seed=111, scope=scope, heads=heads)
    wf = _lib.NativeWavefunction(model, N, 1, (H,))
    wf.set_params(prm, scope=scope)
    opt = Adam()
    T = dict(step=0.0, grad=0.0, adam=0.0, setp=0.0)
    for it in range(100):
        t0 = time.perf_counter()
        m = wf.vmc_step(ns, seed=111, step=it, couplings=coup)["moments"]
        t1 = time.perf_counter()
        mean = complex(m[0] / m[2], m[3] / m[2]) if cplx else m[0] / m[2]
        g = cost_gradient(wf, prm, scope, mean, m[2])
        t2 = time.perf_counter()
        prm = opt.step(prm, g, 5e-4)
        t3 = time.perf_counter()
        wf.set_params(prm, scope=scope)
        t4 = time.perf_counter()
        if it >= 20:
            T["step"] += t1 - t0; T["grad"] += t2 - t1; T["adam"] += t3 - t2; T["setp"] += t4 - t3
    print(model, N, H, ns, {k: round(v / 80 * 1e3, 4) for k, v in T.items()}, "sum", round(sum(T.values()) / 80 * 1e3, 4))
N = 40
run(_lib.MODEL_CRNN_U1, N, 50, 10000, ("wf_dense_ampl", "wf_dense_phase"), np.concatenate([np.ones(N), 0.5 * np.ones(N), np.zeros(N), [0.0, 0.0]]), True)
run(_lib.MODEL_GRU1D, 80, 50, 10000, ("wf_dense",), np.append(np.ones(80), 1.0), False)
