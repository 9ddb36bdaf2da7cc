"""Hot-path part of 2DTFIM_1DRNN/Training1DRNN_2DTFIM.py: Ising2D_local_energies (:13-81)."""
from ..estimators import Ising2D_local_energies  # noqa: F401
from .RNNwavefunction import RNNwavefunction  # noqa: F401
from ..training import run_2DTFIM_1DRNN as run_2DTFIM  # noqa: F401
