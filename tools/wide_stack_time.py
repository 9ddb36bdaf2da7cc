"""Times vmc_step and the gradient for the widths whose weight images are read through L2 - stacked layers above 52 units, single
layers above 100: python tools/wide_stack_time.py"""
import time, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnwavefunctions_amd import _lib, params as P
from rnnwavefunctions_amd.training import cost_gradient
def run(model, N, H, L, ns, heads, couplings):
    prm = P.scale_kernels(P.init_gru_params([H] * L, seed=1, heads=heads), 1.5)
    wf = _lib.NativeWavefunction(model, N, 1, (H,) * L)
    wf.set_params(prm, scope="RNNwavefunction")
    out = wf.vmc_step(ns, seed=1, step=0, couplings=couplings, want_eloc=True)
    t0 = time.perf_counter()
    for it in range(3):
        out = wf.vmc_step(ns, seed=1, step=it + 1, couplings=couplings, want_eloc=True)
    t1 = time.perf_counter()
    m = out["moments"]
    mean = complex(m[0] / m[2], m[3] / m[2]) if model == _lib.MODEL_CRNN_U1 else m[0] / m[2]
    g = cost_gradient(wf, prm, "RNNwavefunction", mean, ns)
    t2 = time.perf_counter()
    for it in range(3):
        g = cost_gradient(wf, prm, "RNNwavefunction", mean, ns)
    t3 = time.perf_counter()
    print("model %d N=%d units=%s ns=%d: vmc_step %.2f ms, gradient %.2f ms, params %d" % (model, N, [H] * L, ns, (t1 - t0) / 3 * 1e3, (t3 - t2) / 3 * 1e3, P.count_params(prm)))
N = 20
run(_lib.MODEL_CRNN_U1, N, 100, 3, 500, ("wf_dense_ampl", "wf_dense_phase"), np.concatenate([np.ones(N), 0.2 * np.ones(N), np.zeros(N), [0.0, 0.0]]))
run(_lib.MODEL_CRNN_U1, N, 50, 3, 500, ("wf_dense_ampl", "wf_dense_phase"), np.concatenate([np.ones(N), 0.2 * np.ones(N), np.zeros(N), [0.0, 0.0]]))
run(_lib.MODEL_GRU1D, N, 100, 2, 500, ("wf_dense",), np.append(np.ones(N), 1.0))
run(_lib.MODEL_GRU1D, N, 64, 2, 500, ("wf_dense",), np.append(np.ones(N), 1.0))
run(_lib.MODEL_GRU1D, N, 50, 2, 500, ("wf_dense",), np.append(np.ones(N), 1.0))
run(_lib.MODEL_GRU1D, 80, 100, 3, 10000, ("wf_dense",), np.append(np.ones(80), 1.0))
for H in (100, 128, 256):
    run(_lib.MODEL_GRU1D, 20, H, 1, 500, ("wf_dense",), np.append(np.ones(20), 1.0))
run(_lib.MODEL_GRU1D, 80, 128, 1, 10000, ("wf_dense",), np.append(np.ones(80), 1.0))
run(_lib.MODEL_GRU1D, 80, 256, 1, 10000, ("wf_dense",), np.append(np.ones(80), 1.0))
run(_lib.MODEL_CRNN_U1, 40, 256, 1, 10000, ("wf_dense_ampl", "wf_dense_phase"), np.concatenate([np.ones(40), 0.5 * np.ones(40), np.zeros(40), [0.0, 0.0]]))
