"""1D GRU over the flattened 2D lattice, float64 (2DTFIM_1DRNN/RNNwavefunction.py:8-130)."""
from ..wavefunctions import GRUWavefunction2DRaster as RNNwavefunction  # noqa: F401
