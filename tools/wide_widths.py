"""VMC step time of the positive RNN at config 2's size (N=80, 10 000 samples) across widths, incl. those above 100 units (f32-input
MFMA, image through L2), and of the complex RNN at config 3's size:  python tools/wide_widths.py [H ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rnnwavefunctions_amd import _lib, params as P
N, ns = 80, 10000
if sys.argv[1:2] == ["stacks"]:
    for units in [(68, 68), (100, 100), (68, 68, 68), (100, 100, 100), (100, 100, 100, 100)]:
        prm = P.init_gru_params(list(units), seed=1)
        wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, N, 1, units)
        wf.set_params(prm, scope="RNNwavefunction")
        c = np.append(np.ones(N), 1.0)
        wf.vmc_step(ns, seed=1, step=0, couplings=c)
        t0 = time.perf_counter()
        for i in range(3):
            wf.vmc_step(ns, seed=1, step=1 + i, couplings=c)
        print("stack units=%-22s engine=%-7s step %8.3f ms" % (units, wf.engine_name(), (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    sys.exit(0)
if sys.argv[1:2] == ["f64"]:
    for H in [36, 52, 68, 84, 100]:
        prm = P.init_gru_params([H], seed=1, dtype=np.float64)
        wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D_F64, 8, 8, (H,))
        wf.set_params(prm, scope="RNNwavefunction")
        c = np.append(np.ones(64), 2.0)
        wf.vmc_step(4000, seed=1, step=0, couplings=c)
        wf.timing_enable(1); wf.timing_reset()
        t0 = time.perf_counter()
        for i in range(3):
            wf.vmc_step(4000, seed=1, step=1 + i, couplings=c)
        dt = (time.perf_counter() - t0) / 3
        flip = wf.timing_get(1)
        cells = 64 * 65 / 2 * 4000
        print("f64 GRU 8x8 H=%3d step %8.3f ms  flip %8.3f ms  %.1f TF/s algorithmic" % (H, dt * 1e3, flip["total_ms"] / max(flip["launches"], 1),
              cells * 6.0 * H * H / (flip["total_ms"] / max(flip["launches"], 1) * 1e-3) / 1e12), flush=True)
    sys.exit(0)
if sys.argv[1:2] == ["grad"]:
    for H in [50, 100, 132, 196, 260]:
        prm = P.init_gru_params([H], seed=1)
        shapes = {k.split("/", 1)[1]: v.shape for k, v in prm.items()}
        wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, N, 1, (H,))
        wf.set_params(prm, scope="RNNwavefunction")
        c = np.append(np.ones(N), 1.0)
        m = wf.vmc_step(ns, seed=1, step=0, couplings=c)["moments"]
        wf.vmc_gradient(m[0] / m[2], m[2], shapes)
        wf.timing_enable(1); wf.timing_reset()
        wf.synchronize(); t0 = time.perf_counter()
        for i in range(3):
            wf.vmc_gradient(m[0] / m[2], m[2], shapes)
        bptt, gemm = wf.timing_get(3), wf.timing_get(4)
        print("gradient H=%3d  %8.3f ms   (BPTT kernel %.3f ms, weight-gradient GEMM %.3f ms per call)" %
              (H, (time.perf_counter() - t0) / 3 * 1e3, bptt["total_ms"] / 3, gemm["total_ms"] / 3), flush=True)
    sys.exit(0)
for H in [int(a) for a in sys.argv[1:]] or [50, 68, 100, 104, 128, 132, 160, 200, 256]:
    prm = P.init_gru_params([H], seed=1)
    wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, N, 1, (H,))
    wf.set_params(prm, scope="RNNwavefunction")
    c = np.append(np.ones(N), 1.0)
    wf.vmc_step(ns, seed=1, step=0, couplings=c)
    wf.timing_enable(1); wf.timing_reset()
    reps = 3 if H > 100 else 10
    t0 = time.perf_counter()
    for i in range(reps):
        wf.vmc_step(ns, seed=1, step=1 + i, couplings=c)
    dt = (time.perf_counter() - t0) / reps
    flip, base = wf.timing_get(1), wf.timing_get(0)
    cells = N * (N + 1) / 2 * ns
    flops = cells * (6.0 * H * H + 12 * H)
    print("H=%3d engine=%-7s step %8.3f ms  flip %8.3f ms  base %7.3f ms   %.1f TF/s algorithmic (flip)" %
          (H, wf.engine_name(), dt * 1e3, flip["total_ms"] / max(flip["launches"], 1), base["total_ms"] / max(base["launches"], 1),
           flops / (flip["total_ms"] / max(flip["launches"], 1) * 1e-3) / 1e12), flush=True)

N, ns = 40, 10000
for H in [int(a) for a in sys.argv[1:]] or [50, 100, 132, 196, 260]:
    prm = P.init_gru_params([H], seed=1, heads=("wf_dense_ampl", "wf_dense_phase"))
    wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,))
    wf.set_params(prm, scope="RNNwavefunction")
    c = np.concatenate([np.ones(N), 0.5 * np.ones(N), np.zeros(N), [0.0, 0.0]])
    wf.vmc_step(ns, seed=1, step=0, couplings=c)
    reps = 3 if H > 100 else 10
    t0 = time.perf_counter()
    for i in range(reps):
        wf.vmc_step(ns, seed=1, step=1 + i, couplings=c)
    dt = (time.perf_counter() - t0) / reps
    print("cRNN H=%3d engine=%-7s step %8.3f ms" % (H, wf.engine_name(), dt * 1e3), flush=True)
