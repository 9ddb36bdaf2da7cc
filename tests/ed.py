"""Independent dense exact diagonalisation used by the known-answer tests.

Written from the Hamiltonians themselves (not from the reference's notebook code); the
notebooks' recorded ground-state energies are the golden numbers
(Tutorials/1DTFIM/Tutorial_1DTFIM.ipynb cell 8: -12.38148999965476;
 Tutorials/J1J2/Tutorial_1DJ1J2.ipynb cell 8: -3.9855798336170905).

Basis state index k <-> configuration conftest.all_configs(N)[k] (site 0 = most significant
bit; 1 = up).
"""
import numpy as np


def _bit(k, site, N):
    return (k >> (N - 1 - site)) & 1


def tfim_hamiltonian(Jz, Bx, N):
    """H = -sum_i Jz_i s^z_i s^z_{i+1} - Bx sum_i s^x_i, open chain, s = Pauli matrices."""
    D = 2 ** N
    H = np.zeros((D, D))
    for k in range(D):
        for i in range(N - 1):
            H[k, k] += -Jz[i] * (1.0 if _bit(k, i, N) == _bit(k, i + 1, N) else -1.0)
        for i in range(N):
            H[k ^ (1 << (N - 1 - i)), k] += -Bx
    return H


def tfim2d_hamiltonian(Jz, Bx, Nx, Ny):
    """Sites (i, j) <-> flat index i*Ny + j; bonds (i,j)-(i+1,j) weighted Jz[i,j] and
    (i,j)-(i,j+1) weighted Jz[i,j] (2DTFIM_2DRNN/Training2DRNN_2DTFIM.py:33-49)."""
    N = Nx * Ny
    D = 2 ** N
    H = np.zeros((D, D))
    for k in range(D):
        for i in range(Nx):
            for j in range(Ny):
                a = _bit(k, i * Ny + j, N)
                if i + 1 < Nx:
                    H[k, k] += -Jz[i, j] * (1.0 if a == _bit(k, (i + 1) * Ny + j, N) else -1.0)
                if j + 1 < Ny:
                    H[k, k] += -Jz[i, j] * (1.0 if a == _bit(k, i * Ny + j + 1, N) else -1.0)
        for s in range(N):
            H[k ^ (1 << (N - 1 - s)), k] += -Bx
    return H


def j1j2_hamiltonian(J1, J2, N, periodic=False, marshall=False):
    """H = sum_i J1_i S_i.S_{i+1} + sum_i J2_i S_i.S_{i+2}, spin-1/2 operators."""
    D = 2 ** N
    H = np.zeros((D, D))
    for dist, J in ((1, J1), (2, J2)):
        lim = N if periodic else N - dist
        for i in range(lim):
            j = (i + dist) % N
            for k in range(D):
                if _bit(k, i, N) == _bit(k, j, N):
                    H[k, k] += 0.25 * J[i]
                else:
                    H[k, k] -= 0.25 * J[i]
                    k2 = k ^ (1 << (N - 1 - i)) ^ (1 << (N - 1 - j))
                    H[k2, k] += (-0.5 if (marshall and dist == 1) else 0.5) * J[i]
    return H
