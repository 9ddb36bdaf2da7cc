"""1DTFIM/TrainingRNN_1DTFIM.py: the estimator Ising_local_energies (:13-75) and the training driver
run_1DTFIM (:79-229; gradient and Adam step of SURVEY.md 8f rows f1/f2, weights saved as .npz)."""
from ..estimators import Ising_local_energies  # noqa: F401
from .RNNwavefunction import RNNwavefunction  # noqa: F401
from ..training import run_1DTFIM  # noqa: F401,E402  (gradient + Adam on the GPU)
