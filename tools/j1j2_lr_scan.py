import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from rnnwavefunctions_amd import training as T
g = np.load(os.path.join("tests", "golden", "notebook_trajectories.npz"))
print("reference:            " + " ".join("%7.3f" % g["j1j2_re"][s // 10] for s in (0, 50, 100, 200, 300, 500, 700, 1000, 1500, 2000, 3000)))
for lr in (5e-4, 7.5e-4, 1e-3, 1.5e-3):
    for seed in (111, 2, 5):
        e, v = T.run_J1J2(numsteps=3000, systemsize=10, J1_=1.0, J2_=0.2, Marshall_sign=False, num_units=10, num_layers=1, numsamples=200, learningrate=lr, seed=seed, verbose=False)
        e = np.real(np.array(e))
        print("lr %.2e seed %3d:  " % (lr, seed) + " ".join("%7.3f" % e[s] for s in (0, 50, 100, 200, 300, 500, 700, 1000, 1500, 2000, 3000)) + "   last100 %.4f" % e[-100:].mean())
