"""2D RNN wave function (2DTFIM_2DRNN/RNNwavefunction.py:5-200)."""
from ..wavefunctions import MDRNNWavefunction2D as RNNwavefunction  # noqa: F401
