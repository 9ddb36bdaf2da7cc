#!/usr/bin/env python3
"""Wall time per bare VMC step (moments only, per-kernel HIP-event timing off, which bench.py needs on):
    python tools/step_latency.py [cfg1|cfg2|...] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg1"])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wf, prm, couplings = bench.make_wavefunction(wl, device=0)
for it in range(10):
    wf.vmc_step(wl["ns"], seed=111, step=it, couplings=couplings)
wf.synchronize()
t0 = time.perf_counter()
for it in range(steps):
    m = wf.vmc_step(wl["ns"], seed=111, step=10 + it, couplings=couplings)["moments"]
wf.synchronize()
dt = (time.perf_counter() - t0) / steps
print("%s  %.4f ms per step  %.4g samples*sites/s  <E> %.6f" % (sys.argv[1] if len(sys.argv) > 1 else "cfg1", dt * 1e3,
                                                                 wl["ns"] * wl["N"] / dt, m[0] / m[2]))
