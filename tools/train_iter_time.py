"""Whole training iterations (sample + local energies + gradient + Adam + re-pack of the weight images) of the drivers at several
sizes, with the iteration resident on the device (rnnwf_train_steps, the default) and with the optimizer on the host:
    python tools/train_iter_time.py"""
import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from rnnwavefunctions_amd import training as T
cases = [("run_1DTFIM", T.run_1DTFIM, dict(systemsize=80, num_units=50, numsamples=10000)),
         ("run_1DTFIM", T.run_1DTFIM, dict(systemsize=20, num_units=50, numsamples=500)),
         ("run_1DTFIM", T.run_1DTFIM, dict(systemsize=10, num_units=10, numsamples=200)),
         ("run_J1J2", T.run_J1J2, dict(systemsize=40, num_units=50, numsamples=10000, J2_=0.5)),
         ("run_J1J2", T.run_J1J2, dict(systemsize=10, num_units=10, numsamples=200, J2_=0.2)),
         ("run_2DTFIM_2DRNN", T.run_2DTFIM_2DRNN, dict(systemsize_x=5, systemsize_y=5, num_units=50, numsamples=500)),      # the run script's size
         ("run_2DTFIM_2DRNN", T.run_2DTFIM_2DRNN, dict(systemsize_x=12, systemsize_y=12, num_units=50, numsamples=10000)),  # config 4's size
         ("run_2DTFIM_1DRNN", T.run_2DTFIM_1DRNN, dict(systemsize_x=5, systemsize_y=5, num_units=50, numsamples=500))]
only = sys.argv[1] if len(sys.argv) > 1 else ""
cases = [c for c in cases if only in c[0]]
for name, run, kw in cases:
    NS = 20 if kw.get("systemsize_x") == 12 else 100
    for mode in (True, False, True, False):
        T.DEVICE_TRAINING = mode
        run(numsteps=9, verbose=False, **kw)
        t0 = time.perf_counter(); e, _ = run(numsteps=NS - 1, verbose=False, **kw); t1 = time.perf_counter()
        print("%s %s %s: %.3f ms per iteration   E[-1] = %s" % (name, kw, "device-resident" if mode else "host optimizer ", (t1 - t0) / NS * 1e3, e[-1]))
