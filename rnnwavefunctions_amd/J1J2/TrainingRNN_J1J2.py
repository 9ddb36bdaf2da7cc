"""Hot-path part of J1J2/TrainingRNN_J1J2.py: J1J2MatrixElements (:12-93), J1J2Slices (:95-127) and the
local-energy assembly (:255-279, here the fused J1J2_local_energies) and the training driver run_J1J2 (:131-308;
gradient of the complex cost and Adam step; weights saved as .npz)."""
from ..estimators import J1J2_local_energies, J1J2MatrixElements, J1J2Slices  # noqa: F401
from .ComplexRNNwavefunction import RNNwavefunction  # noqa: F401
from ..training import run_J1J2  # noqa: F401,E402
