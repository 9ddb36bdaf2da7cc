"""Parity-symmetric variant (1DTFIM/RNNwavefunction_paritysym.py:7-145)."""
from ..wavefunctions import GRUWavefunction1DParity as RNNwavefunction  # noqa: F401
