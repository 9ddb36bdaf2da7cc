"""Our trajectories beside the ones the reference's notebooks record (tests/golden/notebook_trajectories.npz): python tools/notebook_trajectories.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from rnnwavefunctions_amd import training as T
g = np.load(os.path.join("tests", "golden", "notebook_trajectories.npz"))
e, v = T.run_J1J2(numsteps=3000, systemsize=10, J1_=1.0, J2_=0.2, Marshall_sign=False, num_units=10, num_layers=1, numsamples=200, learningrate=5e-4, seed=111, verbose=False)
print("J1J2 step   reference (re, var)        here (re, im, var)")
for s in (0, 10, 20, 50, 100, 200, 300, 500, 700, 1000, 1500, 2000, 2500, 3000):
    i = s // 10
    print("%5d   %9.4f %8.4f      %9.4f %8.4f %8.4f" % (s, g["j1j2_re"][i], g["j1j2_var"][i], np.real(e[s]), np.imag(e[s]), v[s]))
e, v = T.run_1DTFIM(numsteps=1000, systemsize=10, num_units=10, Bx=1, num_layers=1, numsamples=200, learningrate=5e-3, seed=111, verbose=False)
print("TFIM step   reference (E, var)        here (E, var)")
for s in (0, 10, 20, 50, 100, 200, 300, 500, 700, 1000):
    i = s // 10
    print("%5d   %9.4f %8.4f      %9.4f %8.4f" % (s, g["tfim_e"][i], g["tfim_var"][i], e[s], v[s]))
