// util_kernels.h - HBM-bound helpers around the recurrent kernels: spin packing, TFIM local-energy
// assembly, energy moments.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace rnnwf {

// (B, N) int32 row-major  ->  bits[W][B].  Chain position p reads column col_of_pos[p] (identity when
// nullptr); `reverse` reads column N-1-p (RNNwavefunction_paritysym.py:125).
__global__ void pack_bits_kernel(const int32_t* __restrict__ samples, int64_t B, int N,
                                 const int32_t* __restrict__ col_of_pos, int reverse, uint32_t* __restrict__ bits) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = blockIdx.y;
    if (s >= B) return;
    uint32_t word = 0;
    for (int b = 0; b < 32; ++b) {
        const int p = w * 32 + b;
        if (p >= N) break;
        const int col = reverse ? N - 1 - p : (col_of_pos ? col_of_pos[p] : p);
        word |= (uint32_t)(samples[s * N + col] & 1) << b;
    }
    bits[(int64_t)w * B + s] = word;
}

// bits[W][B] -> (B, N) int32 row-major (column `col` shows chain position pos_of_col[col]).
__global__ void unpack_bits_kernel(const uint32_t* __restrict__ bits, int64_t B, int N,
                                   const int32_t* __restrict__ pos_of_col, int32_t* __restrict__ samples) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N) return;
    const int64_t s = idx / N;
    const int col = (int)(idx - s * N);
    const int p = pos_of_col ? pos_of_col[col] : col;
    samples[idx] = (bits[(int64_t)(p >> 5) * B + s] >> (p & 31)) & 1;
}

// E_loc of the transverse-field Ising model on an Nx x Ny open lattice (1D chain: Nx = 1):
//   1DTFIM/TrainingRNN_1DTFIM.py:31-38,70-74 ; 2DTFIM_2DRNN/Training2DRNN_2DTFIM.py:33-49,78-81.
// Site (i, j) = flat k = i*Ny + j sits at chain position pos_of_site[k]; Jz is (Nx, Ny) row-major.
// Block = 32 samples x 8 site groups: group g sums the off-diagonal terms of sites g, g+8, ... (one f64 exp each),
// group 0 also the bond terms; the eight partial sums are added in fixed order (reproducible).
constexpr int kElocSamples = 32, kElocGroups = 8;
__global__ void __launch_bounds__(kElocSamples * kElocGroups)
tfim_eloc_kernel(const uint32_t* __restrict__ bits, const double* __restrict__ lpq, int64_t ns, int Nx, int Ny,
                 const int32_t* __restrict__ pos_of_site, const double* __restrict__ Jz, double Bx,
                 double* __restrict__ eloc) {
    __shared__ double part[kElocGroups][kElocSamples];
    const int ls = threadIdx.x % kElocSamples, g = threadIdx.x / kElocSamples;
    const int64_t s = (int64_t)blockIdx.x * kElocSamples + ls;
    const bool valid = s < ns;
    const int64_t sc = valid ? s : ns - 1;
    auto spin = [&](int k) {
        const int p = pos_of_site ? pos_of_site[k] : k;
        return (int)((bits[(int64_t)(p >> 5) * ns + sc] >> (p & 31)) & 1);
    };
    double e = 0.0;
    if (g == 0) {
        for (int i = 0; i + 1 < Nx; ++i)
            for (int j = 0; j < Ny; ++j)
                e += (spin(i * Ny + j) == spin((i + 1) * Ny + j) ? 1.0 : -1.0) * (-Jz[i * Ny + j]);
        for (int j = 0; j + 1 < Ny; ++j)
            for (int i = 0; i < Nx; ++i)
                e += (spin(i * Ny + j) == spin(i * Ny + j + 1) ? 1.0 : -1.0) * (-Jz[i * Ny + j]);
    }
    if (Bx != 0.0) {
        const int N = Nx * Ny;
        const double l0 = 0.5 * lpq[sc];
        double acc = 0.0;
        for (int k = g; k < N; k += kElocGroups) acc += exp(0.5 * lpq[(int64_t)(k + 1) * ns + sc] - l0);
        e += -Bx * acc;
    }
    part[g][ls] = e;
    __syncthreads();
    if (g == 0 && valid) {
        double t = part[0][ls];
#pragma unroll
        for (int k = 1; k < kElocGroups; ++k) t += part[k][ls];
        eloc[s] = t;
    }
}

// moments[0..3] = { sum Re E, sum (Re E)^2, n, sum Im E }; one workgroup, fixed order -> reproducible.
// re/im may be f64 (TFIM, im == nullptr) or interleaved f32 pairs (J1J2).
template <typename TE>
__global__ void __launch_bounds__(1024) moments_kernel(const TE* __restrict__ e, int64_t ns, int stride,
                                                       int has_im, double* __restrict__ moments, double* host_copy) {
    __shared__ double sh[3][1024];
    double s1 = 0.0, s2 = 0.0, si = 0.0;
    for (int64_t s = threadIdx.x; s < ns; s += blockDim.x) {
        const double re = (double)e[s * stride];
        s1 += re;
        s2 += re * re;
        if (has_im) si += (double)e[s * stride + 1];
    }
    sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = s2; sh[2][threadIdx.x] = si;
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + w];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + w];
            sh[2][threadIdx.x] += sh[2][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        moments[0] = sh[0][0]; moments[1] = sh[1][0]; moments[2] = (double)ns; moments[3] = sh[2][0];
        if (host_copy) { host_copy[0] = sh[0][0]; host_copy[1] = sh[1][0]; host_copy[2] = (double)ns; host_copy[3] = sh[2][0]; }
    }
}

// parity-symmetric combination log(0.5 (exp(a) + exp(b)))  (RNNwavefunction_paritysym.py:145)
__global__ void parity_combine_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                      double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = log(0.5 * (exp(a[i]) + exp(b[i])));
}

// Parity-symmetric model, gradient: log P_sym = log(0.5 (P_F + P_R)) gives  d log P_sym = aF d log P_F + aR d log P_R  with the shares
// aF = P_F / (P_F + P_R), aR = 1 - aF.  In place: lpF <- aF, lpR <- aR.
__global__ void parity_share_kernel(double* __restrict__ lpF, double* __restrict__ lpR, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double aF = 1.0 / (1.0 + exp(lpR[i] - lpF[i]));
    lpF[i] = aF;
    lpR[i] = 1.0 - aF;
}

}  // namespace rnnwf
