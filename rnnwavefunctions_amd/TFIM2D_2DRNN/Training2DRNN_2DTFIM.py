"""Hot-path part of 2DTFIM_2DRNN/Training2DRNN_2DTFIM.py: Ising2D_local_energies (:13-83)."""
from ..estimators import Ising2D_local_energies  # noqa: F401
from .MDRNNcell import MDRNNcell  # noqa: F401
from .RNNwavefunction import RNNwavefunction  # noqa: F401
from ..training import run_2DTFIM_2DRNN as run_2DTFIM  # noqa: F401
