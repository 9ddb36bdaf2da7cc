"""Hot-path part of 1DTFIM/TrainingRNN_1DTFIM.py: the estimator (:13-75).  The training driver
(run_1DTFIM, :79-229: Adam, autodiff, checkpoints) is outside the scope of this build."""
from ..estimators import Ising_local_energies  # noqa: F401
from .RNNwavefunction import RNNwavefunction  # noqa: F401
