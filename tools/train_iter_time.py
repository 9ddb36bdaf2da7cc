"""Whole training iterations (sample + local energies + gradient + Adam + parameter upload) of run_1DTFIM / run_J1J2 at four sizes: python tools/train_iter_time.py"""
import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from rnnwavefunctions_amd.training import run_1DTFIM, run_J1J2
for kw in (dict(systemsize=80, num_units=50, numsamples=10000), dict(systemsize=20, num_units=50, numsamples=500), dict(systemsize=10, num_units=10, numsamples=200)):
    run_1DTFIM(numsteps=5, verbose=False, **kw)
    t0 = time.perf_counter(); run_1DTFIM(numsteps=60, verbose=False, **kw); t1 = time.perf_counter()
    print("run_1DTFIM", kw, "%.3f ms per iteration" % ((t1 - t0) / 61 * 1e3))
kw = dict(systemsize=40, num_units=50, numsamples=10000, J2_=0.5)
run_J1J2(numsteps=5, verbose=False, **kw)
t0 = time.perf_counter(); run_J1J2(numsteps=60, verbose=False, **kw); t1 = time.perf_counter()
print("run_J1J2", kw, "%.3f ms per iteration" % ((t1 - t0) / 61 * 1e3))
