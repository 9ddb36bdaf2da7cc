#!/usr/bin/env python3
"""Soak: the same VMC step (same seed / step) repeated many times must return bit-identical moments - any difference is a
scheduling hazard or a race (atomically compacted work lists may change ORDER between runs, results may not).
    python tools/soak_determinism.py [cfg2 300] [cfg3 300] [cfg5 12] ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402

args = sys.argv[1:] or ["cfg2", "300", "cfg3", "300", "cfg1", "500", "cfg5", "12", "cfg4", "30"]
bad = 0
for name, reps in zip(args[0::2], args[1::2]):
    wl = dict(bench.WORKLOADS[name])
    wf, prm, couplings = bench.make_wavefunction(wl, device=0)
    ref = wf.vmc_step(wl["ns"], seed=111, step=3, couplings=couplings)["moments"]
    diff = 0
    for it in range(int(reps)):
        if it % 3 == 1:                       # interleave other steps: buffers get overwritten in between
            wf.vmc_step(wl["ns"], seed=111, step=4 + it, couplings=couplings)
        m = wf.vmc_step(wl["ns"], seed=111, step=3, couplings=couplings)["moments"]
        diff += int(not np.array_equal(m, ref))
    print("%-5s %4d repetitions, engine %-7s: %d differing  (<E> %.9f)" % (name, int(reps), wf.engine_name(), diff, ref[0] / ref[2]))
    bad += diff
print("soak done, differing steps:", bad)
sys.exit(1 if bad else 0)
