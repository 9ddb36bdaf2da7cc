import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from rnnwavefunctions_amd import training as T
for seed in (111, 1, 2, 3, 4, 5, 6, 7):
    e, v = T.run_J1J2(numsteps=3000, systemsize=10, J1_=1.0, J2_=0.2, Marshall_sign=False, num_units=10, num_layers=1, numsamples=200, learningrate=5e-4, seed=seed, verbose=False)
    e = np.real(np.array(e))
    print("seed %3d: E0 %.3f  E200 %.3f  E500 %.3f  E1000 %.3f  E2000 %.3f  last100 %.4f  var %.4f" % (seed, e[0], e[200], e[500], e[1000], e[2000], e[-100:].mean(), np.mean(v[-100:])))
