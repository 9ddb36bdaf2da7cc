"""``from RNNwavefunction import RNNwavefunction`` of 1DTFIM/ (1DTFIM/RNNwavefunction.py:7-118)."""
from ..wavefunctions import GRUWavefunction1D as RNNwavefunction  # noqa: F401
