"""CPU oracle for the RNN-wavefunction VMC hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in dtype-faithful NumPy (and plain C under ``oracle/c``),
the algorithm of the reference's ``RNNwavefunction.sample`` / ``log_probability`` /
``log_amplitude`` and of its TFIM / J1J2 local-energy estimators.  It exists so
that the hand-written HIP path can be checked against something; it is never
shipped and never measured as the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``rnnwavefunctions_amd/`` imports it.

Pinning status:
  * estimators (``Ising_local_energies``, ``Ising2D_local_energies``,
    ``J1J2MatrixElements``, ``J1J2Slices``): PINNED against the reference's own
    NumPy code, run in the build container, via ``tests/golden/*.npz``
    (generator: ``tests/golden/make_fixtures.py``).
  * Hamiltonian conventions: PINNED by the exact-diagonalisation energies the
    reference's notebooks record (-12.38148999965476, -3.9855798336170905).
  * cell structure: PINNED by the parameter counts 422 / 444 in the notebooks.
  * TensorFlow-1.13 kernel arithmetic (GRU/softmax/multinomial bit patterns,
    Philox stream, glorot draws): PARITY UNPINNED - TensorFlow is not installable
    here and the reference holds no fixture at that boundary.  The restatement
    follows the published TF 1.13.1 semantics and is checked by exact
    mathematical identities (normalisation, psi^T H psi, zero magnetisation).
"""
