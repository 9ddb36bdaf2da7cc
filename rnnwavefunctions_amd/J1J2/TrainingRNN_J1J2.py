"""Hot-path part of J1J2/TrainingRNN_J1J2.py: J1J2MatrixElements (:12-93), J1J2Slices (:95-127) and the
local-energy assembly (:255-279, here the fused J1J2_local_energies).  run_J1J2 itself (optimizer,
checkpoints) is outside the scope of this build."""
from ..estimators import J1J2_local_energies, J1J2MatrixElements, J1J2Slices  # noqa: F401
from .ComplexRNNwavefunction import RNNwavefunction  # noqa: F401
