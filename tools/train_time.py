#!/usr/bin/env python3
"""Wall time of one training iteration (vmc_step / gradient / Adam + parameter upload) at a few sizes:
   python tools/train_time.py            (on the MI355X box)"""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnwavefunctions_amd import _lib, params as P                      # noqa: E402
from rnnwavefunctions_amd.training import Adam, cost_gradient           # noqa: E402

for name, N, H, ns in (("cfg2 (N=80, 50 units, 10000 samples)", 80, 50, 10000),
                       ("run_1dTFIM.py (N=20, 50 units, 500 samples)", 20, 50, 500),
                       ("notebook (N=10, 10 units, 200 samples)", 10, 10, 200)):
    prm = P.init_gru_params([H], seed=111)
    wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, N, 1, (H,))
    wf.set_params(prm, scope="RNNwavefunction")
    c = np.append(np.ones(N), 1.0)
    opt = Adam()
    tv = tg = ta = tu = 0.0
    iters, warm = 30, 5
    for it in range(iters + warm):
        a = time.perf_counter()
        m = wf.vmc_step(ns, 111, it, c)["moments"]
        b = time.perf_counter()
        g = cost_gradient(wf, prm, "RNNwavefunction", m[0] / m[2], ns)
        c2 = time.perf_counter()
        prm = opt.step(prm, g, 5e-3)
        d = time.perf_counter()
        wf.set_params(prm, scope="RNNwavefunction")
        e = time.perf_counter()
        if it >= warm:
            tv += b - a; tg += c2 - b; ta += d - c2; tu += e - d
    k = 1e3 / iters
    print("%-46s vmc_step %.3f ms  gradient %.3f ms  Adam %.3f ms  set_params %.3f ms  | total %.3f ms  E=%.3f" %
          (name, tv * k, tg * k, ta * k, tu * k, (tv + tg + ta + tu) * k, m[0] / m[2]))
