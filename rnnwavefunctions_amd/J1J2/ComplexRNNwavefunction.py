"""``from ComplexRNNwavefunction import RNNwavefunction`` (J1J2/ComplexRNNwavefunction.py:15-169)."""
from ..wavefunctions import ComplexGRUWavefunction1D as RNNwavefunction  # noqa: F401
