"""Gradient time of the wide complex RNN and float64 GRU shapes (BPTT kernel / weight-gradient product):  python tools/wide_gradients.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rnnwavefunctions_amd import _lib, params as P
def run(model, nx, ny, H, ns, c, heads, dt):
    prm = P.init_gru_params([H], seed=1, heads=heads, dtype=dt)
    shapes = {k.split("/", 1)[1]: v.shape for k, v in prm.items()}
    wf = _lib.NativeWavefunction(model, nx, ny, (H,))
    wf.set_params(prm, scope="RNNwavefunction")
    m = wf.vmc_step(ns, seed=1, step=0, couplings=c)["moments"]
    me = complex(m[0] / m[2], m[3] / m[2]) if model == _lib.MODEL_CRNN_U1 else m[0] / m[2]
    wf.vmc_gradient(me, m[2], shapes)
    wf.timing_enable(1); wf.timing_reset(); wf.synchronize(); t0 = time.perf_counter()
    for i in range(3): g = wf.vmc_gradient(me, m[2], shapes)
    b, gm = wf.timing_get(3), wf.timing_get(4)
    print("%s H=%3d gradient %.3f ms (BPTT %.3f, GEMM %.3f)  |g| %.6e" % ({_lib.MODEL_CRNN_U1: "cRNN N=40 ns=10000", _lib.MODEL_GRU1D_F64: "f64 GRU 8x8 ns=4000"}[model], H,
          (time.perf_counter() - t0) / 3 * 1e3, b["total_ms"] / 3, gm["total_ms"] / 3, np.sqrt(sum(float((v * v).sum()) for v in g.values()))), flush=True)
N = 40
cc = np.concatenate([np.ones(N), 0.5 * np.ones(N), np.zeros(N), [0.0, 0.0]])
for H in (100, 132, 196, 260): run(_lib.MODEL_CRNN_U1, N, 1, H, 10000, cc, ("wf_dense_ampl", "wf_dense_phase"), np.float32)
for H in (68, 100): run(_lib.MODEL_GRU1D_F64, 8, 8, H, 4000, np.append(np.ones(64), 2.0), ("wf_dense",), np.float64)
