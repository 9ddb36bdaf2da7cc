"""rnnwavefunctions_amd - MI355X-native VMC inner loop for RNN wave functions.

Python host code -> ctypes -> C-ABI (include/rnnwf.h) -> hand-written HIP kernels for
gfx950.  The sub-packages mirror the reference's folders and keep its call signatures:

    TFIM1D        <- 1DTFIM/        RNNwavefunction, RNNwavefunction_paritysym, Ising_local_energies
    J1J2          <- J1J2/          ComplexRNNwavefunction, J1J2MatrixElements, J1J2Slices
    TFIM2D_2DRNN  <- 2DTFIM_2DRNN/  MDRNNcell, RNNwavefunction, Ising2D_local_energies
    TFIM2D_1DRNN  <- 2DTFIM_1DRNN/  RNNwavefunction, Ising2D_local_energies

There is no CPU fallback: importing a wave function without the built HIP library raises.
"""
__version__ = "0.1.0"
