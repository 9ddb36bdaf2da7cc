"""Two hundred training iterations at the reference run script's size (N=20, 50 units, 500 samples) - run under rocprofv3 --kernel-trace --stats for the per-kernel split of an iteration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnwavefunctions_amd import training as T
kw = dict(systemsize=20, num_units=50, numsamples=500)
T.run_1DTFIM(numsteps=9, verbose=False, **kw)
T.run_1DTFIM(numsteps=199, verbose=False, **kw)
