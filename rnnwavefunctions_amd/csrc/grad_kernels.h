// grad_kernels.h - gradient of the VMC cost for the positive GRU RNN.
//
// Reference: cost = mean(log P * E_loc) - mean(E_loc) * mean(log P), differentiated by TensorFlow
// (1DTFIM/TrainingRNN_1DTFIM.py:151-166), i.e.  grad = sum_s w_s dlog P(s)/dtheta,  w_s = (E_s - mean E) / ns.
//
//   gru_bwd_kernel  : back-propagation through time for 16 chains per wave, site N-1 down to 0.  Each site
//                     re-computes its gates from the stored input state (hck, written by the base pass), forms
//                     the pre-activation gradients (VALU) and propagates dL/dh through the transposed weights
//                     on the f32 MFMA (same lane layout trick as the forward: the C/D fragment is the next B
//                     operand).  It writes, per (site, chain), the row  P = [d a_r | d a_u | d q | d y]  and the
//                     row  Q = [h_in | x one-hot | 1]; head gradients are summed in registers, one row per wave,
//                     and head_reduce_kernel adds the rows.
//   tn_gemm_kernel  : dW = P^T Q over all R = N*ns rows (tall-skinny TN GEMM on the f32 / f64 MFMA; operands are read
//                     straight from the row-major arrays, which already have the A/B fragment shape); every block
//                     stores its partial fragments, tn_reduce_kernel adds them in a fixed order.
#pragma once
#include "gru_core.h"

namespace rnnwf {

template <int NFULL, typename T = float>
struct GradLayout {
    static constexpr int KT = 4 * NFULL + 1;
    static constexpr int NT = 3 * NFULL + 1;
    static constexpr int NTO = NFULL + 1;                 // 16-row tiles covering the hidden units
    static constexpr int PCOLS = 16 * (NT + NTO);         // [gate rows in image order | dy in unit-fragment order]
    static constexpr int QCOLS = 16 * NTO;                // [h_in in unit-fragment order ; x0, x1, 1 in the spare slots]
    static constexpr int KB = 3 * KT;                     // k-steps of the backward product (gate g, kt)
    static constexpr int VW = 16 / (int)sizeof(T);        // k-steps per 16-byte LDS vector
    static constexpr int KBG = (KB + VW - 1) / VW;        // groups of VW k-steps (b128 LDS reads)
    static constexpr size_t BWD_BYTES = (size_t)NTO * KBG * 64 * 16;   // [NTO][KBG][64] x 16 B
    static constexpr int HEAD_ROW = 4 * KT + 4;           // per head row: gradient per unit slot (4 KT) + bias (+ pad)
};
// Forward + backward image beyond the 160 KB of LDS (f32 above 68 units, f64 above 52): the backward operand stays in
// global memory and its fragments are read through L2 with buffer loads, one k-group ahead (28 MFMAs ~ 900 cycles).
template <typename T, int NFULL, int NOUT>
struct GradStream {
    static constexpr bool value = GruLayout<T, NFULL, NOUT>::BYTES + GradLayout<NFULL, T>::BWD_BYTES > 160 * 1024;
};

struct GradArgs {
    const void* wimg;          // forward image (GruLayout<float, NFULL, 1>)
    const void* wbwd;          // backward A fragments (GradLayout::BWD_BYTES)
    int32_t N;
    int64_t ns, nsb;
    const uint32_t* bits;
    const void* hck;           // [N-1][nsb][KT][64] T, state after each site
    const double* eloc;        // [ns] f64 (positive RNN) ...
    const float2* eloc_c;      // ... or [ns] complex64 (complex RNN)
    double mean_e, mean_im, inv_norm;   // w_s = (E_s - mean) * inv_norm  (real and imaginary part separately)
    const double* mom;         // device-resident training (train.hip): the step's moments {sum Re E, sum (Re E)^2, n, sum Im E} still on the
                               // device - then mean_e = mom[0] / mom[2], mean_im = mom[3] / mom[2] and inv_norm (the cost's factor 1 or 2) is
                               // divided by mom[2]: the divisions the host path does in Python, in the same arithmetic; nullptr: the three fields above
    const double* wfac;        // [ns] extra factor of w_s or nullptr (parity-symmetric model: the direction's share of P_sym)
    void* P;                   // [N*ns][PCOLS] T
    void* Q;                   // [N*ns][QCOLS] T
    void* head_grad;           // [NOUT][HEAD_ROW] T, zeroed before the launch (written by head_reduce_kernel)
    void* head_part;           // [waves of the grid][NOUT][HEAD_ROW] T: every wave's head-row sums
    // stacked layers: this kernel is then the LAST pass (layer 0); dL/dh of every site arrives from
    // the layer above instead of from the head, and hck holds hck_nl layers per (site, block)
    const void* dh_in;         // [N][nsb][KT][64] T or nullptr
    int32_t hck_nl;            // layers per checkpoint entry (0 or 1: single layer)
};

// Head-row sums of one wave over its chains, stored (not added) to row = part[global wave][NOUT][HEAD_ROW]: slot 4 k + q is a unit
// slot, 4 KT the bias, the rest zero.  head_reduce_kernel adds the waves' rows in a fixed order: no atomics, the gradient is
// bit-reproducible.
template <typename T, int NOUT, int KT>
__device__ __forceinline__ void store_head_part(T* row, int head_row, const T (&hg)[NOUT][KT], const T (&gb)[NOUT], int c, int q) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            T v = hg[o][k];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (c == 0) row[o * head_row + 4 * k + q] = v;
        }
        T v = gb[o];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        if (c == 0) row[o * head_row + 4 * KT + q] = q == 0 ? v : T(0);
    }
}

// out[j] += sum over the nparts rows of part[.][j], j < n; 64 entries per workgroup of 16 waves (wave w adds the rows w, w + 16, ...).
template <typename T>
__global__ void __launch_bounds__(1024) head_reduce_kernel(const T* __restrict__ part, int nparts, int n, T* __restrict__ out) {
    __shared__ T sums[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    T v = T(0);
    if (j < n)
        for (int p = w; p < nparts; p += 16) v += part[(size_t)p * n + j];
    sums[w][lane] = v;
    __syncthreads();
    if (w != 0 || j >= n) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) v += sums[k][lane];
    out[j] += v;
}

// NOUT = 1: positive RNN, L = sum_s w_s log P(s).
// NOUT = 3: complex RNN, L = sum_s [w_re Re log psi(s) + w_im Im log psi(s)]  (J1J2/TrainingRNN_J1J2.py:197:
//           cost = 2 Re(mean(conj(log psi) E) - conj(mean log psi) mean E); the factor 2 is in inv_norm).
// PAIR (small batches: fewer blocks of 16 chains than SIMD pairs): two waves per block.  Wave F re-runs the forward step of site n - 1
// (state from the checkpoint, gates, head) while wave B takes site n's derivatives and the transposed products - the two halves of an
// iteration do not depend on each other, only B's chain dL/dh_n -> dL/dh_{n-1} is sequential.  F hands (h, h', r, u, c, q, z) to B
// through LDS, two workgroup barriers per site.  At the reference run script's size (N = 20, 50 units, 500 samples = 32 blocks) this
// kernel is half of a training iteration's device time and every site costs a lone wave 15 000 cycles (profiles/r04_y_small_iteration.txt).
template <typename T, int NFULL, int NOUT>
struct GradPair {
    static constexpr int KT = GruLayout<T, NFULL, NOUT>::KT;
    static constexpr int NB = 2;                                                   // blocks per workgroup (four waves, one per SIMD)
    static constexpr size_t SLOT = ((size_t)(6 * KT + NOUT) * 64 * sizeof(T) + 15) / 16 * 16;
    static constexpr size_t IMG = GruLayout<T, NFULL, NOUT>::BYTES + GradLayout<NFULL, T>::BWD_BYTES;
    static constexpr size_t LDS = IMG + NB * SLOT;
    static constexpr bool FITS = !GruLayout<T, NFULL, NOUT>::SPILL && LDS <= 160 * 1024;
};

template <typename T, int NFULL, int WAVES, int NOUT, bool PAIR = false>
__global__ void __launch_bounds__(WAVES * 64) gru_bwd_kernel(GradArgs a) {
    using C = GruCore<T, NFULL, NOUT>;
    using G = GradLayout<NFULL, T>;
    using V4 = typename C::V4;
    using VA = typename C::VA;
    constexpr int VW = G::VW;
    constexpr int KT = C::KT, NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool STREAM = GradStream<T, NFULL, NOUT>::value;
    const char* img = C::stage(lds, a.wimg);       // LDS, or the global image where it exceeds LDS (GruLayout::SPILL)
    if constexpr (!STREAM) {
        char* lb = lds + C::L::BYTES;
        const uint4* src = reinterpret_cast<const uint4*>(a.wbwd);
        uint4* dst = reinterpret_cast<uint4*>(lb);
        for (int i = threadIdx.x; i < (int)(G::BWD_BYTES / 16); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const char* lbwd = lds + C::L::BYTES;
    const __amdgpu_buffer_rsrc_t gbwd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wbwd), 0, (int)G::BWD_BYTES, 0x00020000);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    auto bwd_frag = [&](int t, int kg) -> VA {          // fragment (tile t, k-group kg) of the backward operand
        if constexpr (STREAM) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(gbwd, (threadIdx.x & 63) * 16, (t * G::KBG + kg) * 64 * 16, 0);
            return __builtin_bit_cast(VA, v);
        } else {
            return (reinterpret_cast<const VA*>(lbwd) + (threadIdx.x & 63))[(t * G::KBG + kg) * 64];
        }
    };
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    // PAIR: wave 2 b is block slot b's forward wave, wave 2 b + 1 its backward wave; both walk the same blocks
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool fwd_role = !PAIR || (wave & 1) == 0, bwd_role = !PAIR || (wave & 1) == 1;
    constexpr int SLOTS = PAIR ? WAVES / 2 : WAVES;
    const int64_t gw = (int64_t)blockIdx.x * SLOTS + (PAIR ? wave >> 1 : wave);
    const int64_t nw = (int64_t)gridDim.x * SLOTS;
    T* xch = reinterpret_cast<T*>(lds + GradPair<T, NFULL, NOUT>::IMG + (size_t)(wave >> 1) * GradPair<T, NFULL, NOUT>::SLOT) + lane;
    const int N = a.N;
    const T* wd = reinterpret_cast<const T*>(img + C::L::OFF_WD) + q * C::L::WD_Q;
    const int hck_nl = a.hck_nl > 1 ? a.hck_nl : 1;
    T hg[NOUT][KT], gb[NOUT];                             // head-row sums over all chains of this wave
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        gb[o] = T(0);
#pragma unroll
        for (int k = 0; k < KT; ++k) hg[o][k] = T(0);
    }
    for (int64_t base = 0; base < a.nsb; base += nw) {
        // PAIR: every wave of the workgroup runs the same number of barriers - a slot without a block of its own walks the last block
        // again with all its chains invalid (weights 0, nothing stored)
        const bool active = base + gw < a.nsb;
        if (!PAIR && !active) break;
        const int64_t sb = active ? base + gw : a.nsb - 1;
        const int64_t s = sb * kChains + c;
        const bool valid = active && s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        T w = T(0), w_im = T(0);
        if (valid) {
            const double mean_e = a.mom ? a.mom[0] / a.mom[2] : a.mean_e, mean_im = a.mom ? a.mom[3] / a.mom[2] : a.mean_im;
            const double inv_norm = a.mom ? a.inv_norm / a.mom[2] : a.inv_norm;
            if constexpr (NOUT == 1) {
                w = (T)((a.eloc[sc] - mean_e) * inv_norm * (a.wfac ? a.wfac[sc] : 1.0));
            } else {
                const float2 e = a.eloc_c[sc];
                w = (T)(((double)e.x - mean_e) * inv_norm);
                w_im = (T)(((double)e.y - mean_im) * inv_norm);
            }
        }
        auto word = [&](int wi) { return a.bits[(int64_t)wi * a.ns + sc]; };
        // nothing a site needs is fetched in that site: the spin words of sites n and n - 1 sit in registers (the word below is
        // fetched 32 sites ahead) and the input state of site n - 1 is fetched while site n is worked on
        uint32_t wcur = word((N - 1) >> 5), wlow = N > 32 ? word(((N - 1) >> 5) - 1) : 0u;
        auto fetch_state = [&](int n, T (&dst)[KT]) {           // h_in of site n = state after site n - 1 (zero at n = 0)
            if (n > 0) {
                const T* src = reinterpret_cast<const T*>(a.hck) + (((int64_t)(n - 1) * a.nsb + sb) * hck_nl * KT) * 64 + lane;
#pragma unroll
                for (int k = 0; k < KT; ++k) dst[k] = src[k * 64];
            } else {
#pragma unroll
                for (int k = 0; k < KT; ++k) dst[k] = T(0);
            }
        };
        T hpf[KT];
        if (fwd_role) fetch_state(N - 1, hpf);
        T dh[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) dh[k] = T(0);
        int num_up = 0;                                   // complex RNN: up spins among sites < n (for the U(1) mask)
        if constexpr (NOUT == 3)
            for (int wi = 0; wi < (N + 31) / 32; ++wi)
                num_up += __popc(word(wi) & (32 * wi + 32 <= N ? 0xffffffffu : (1u << (N & 31)) - 1u));
        for (int n = N - 1; n >= 0; --n) {
            T h[KT], hn[KT], rg[KT], ug[KT], cc[KT], qv[KT];
            if (fwd_role) {
#pragma unroll
                for (int k = 0; k < KT; ++k) h[k] = hpf[k];
                if (n > 0) fetch_state(n - 1, hpf);
            }
            const int sig = (int)((wcur >> (n & 31)) & 1);
            const int sig_in = n == 0 ? -1 : (n & 31) ? (int)((wcur >> ((n - 1) & 31)) & 1) : (int)(wlow >> 31);
            if ((n & 31) == 0 && n > 0) {
                wcur = wlow;
                wlow = n >= 64 ? word((n >> 5) - 2) : 0u;
            }
            T z[NOUT];
            if (fwd_role) {
                C::step_keep(img, sig_in, h, hn, rg, ug, cc, qv, lane);
                C::head(img, hn, lane, z);
            }
            if constexpr (PAIR) {                                      // F -> B through the block's LDS slot
                if (fwd_role) {
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        xch[(0 * KT + k) * 64] = h[k];  xch[(1 * KT + k) * 64] = hn[k]; xch[(2 * KT + k) * 64] = rg[k];
                        xch[(3 * KT + k) * 64] = ug[k]; xch[(4 * KT + k) * 64] = cc[k]; xch[(5 * KT + k) * 64] = qv[k];
                    }
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) xch[(6 * KT + o) * 64] = z[o];
                }
                __syncthreads();
                if (bwd_role) {
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        h[k] = xch[(0 * KT + k) * 64];  hn[k] = xch[(1 * KT + k) * 64]; rg[k] = xch[(2 * KT + k) * 64];
                        ug[k] = xch[(3 * KT + k) * 64]; cc[k] = xch[(4 * KT + k) * 64]; qv[k] = xch[(5 * KT + k) * 64];
                    }
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) z[o] = xch[(6 * KT + o) * 64];
                }
                __syncthreads();
                if (!bwd_role) continue;                               // the forward wave is done with this site
            }
            // gradient of this site's term w.r.t. the head rows
            T g[NOUT];
            const T p1 = T(1) - prob0(z[0]);
            if constexpr (NOUT == 1) {
                g[0] = w * ((T)sig - p1);                              // d log p(sig) / d(z1 - z0) = sig - p1
            } else {
                num_up -= sig;                                          // ups among sites < n
                bool both = true;                                       // mask: a value that is forced has amplitude 1
                if (2 * n >= N) {
                    const int base = N / 2 - 1;
                    both = (base - (n - num_up) >= 0) && (base - num_up >= 0);
                }
                g[0] = both ? T(0.5) * w * ((T)sig - p1) : T(0);      // d log a(sig) = 1/2 d log p(sig)
                const T zs = sig ? z[2] : z[1];
                const T den = T(1) + (zs < T(0) ? -zs : zs);
                const T gp = w_im * T(3.14159265358979323846) / (den * den);   // d (pi softsign(z)) / dz
                g[1] = sig ? T(0) : gp;
                g[2] = sig ? gp : T(0);
            }
            if (a.dh_in) {                                              // stack: the head sits on the top layer
#pragma unroll
                for (int o = 0; o < NOUT; ++o) g[o] = T(0);
            }
            T dp[VW * G::KBG];
            T dy[KT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) gb[o] += g[o];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                T d = dh[k];                                        // total dL/dh_n of this lane's unit
                if (a.dh_in) d += reinterpret_cast<const T*>(a.dh_in)[(((int64_t)n * a.nsb + sb) * KT + k) * 64 + lane];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    hg[o][k] += g[o] * hn[k];
                    d += g[o] * wd[k * NOUT + o];
                }
                const T du = d * (h[k] - cc[k]);
                const T dc = d * (T(1) - ug[k]);
                dh[k] = d * ug[k];                                      // direct path to h_{n-1}
                dy[k] = dc * (T(1) - cc[k] * cc[k]);
                dp[2 * KT + k] = dy[k] * rg[k];                         // d q
                dp[k] = dy[k] * qv[k] * rg[k] * (T(1) - rg[k]);         // d a_r
                dp[KT + k] = du * ug[k] * (T(1) - ug[k]);               // d a_u
            }
#pragma unroll
            for (int k = G::KB; k < VW * G::KBG; ++k) dp[k] = T(0);
            if (valid) {
                T* prow = reinterpret_cast<T*>(a.P) + ((int64_t)n * a.ns + s) * G::PCOLS + 4 * q;
#pragma unroll
                for (int m = 0; m < NFULL; ++m) {
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) {
                        V4 v = {dp[gt * KT + 4 * m], dp[gt * KT + 4 * m + 1], dp[gt * KT + 4 * m + 2], dp[gt * KT + 4 * m + 3]};
                        *reinterpret_cast<V4*>(prow + (gt * NFULL + m) * 16) = v;
                    }
                    V4 vy = {dy[4 * m], dy[4 * m + 1], dy[4 * m + 2], dy[4 * m + 3]};
                    *reinterpret_cast<V4*>(prow + (NT + m) * 16) = vy;
                }
                V4 vm = {dp[KT - 1], dp[2 * KT - 1], dp[3 * KT - 1], T(0)};    // mixed tile: row 4q + gate
                *reinterpret_cast<V4*>(prow + (NT - 1) * 16) = vm;
                V4 vym = {dy[KT - 1], T(0), T(0), T(0)};
                *reinterpret_cast<V4*>(prow + (NT + NFULL) * 16) = vym;
                T* qrow = reinterpret_cast<T*>(a.Q) + ((int64_t)n * a.ns + s) * G::QCOLS + 4 * q;
#pragma unroll
                for (int m = 0; m < NFULL; ++m) {
                    V4 v = {h[4 * m], h[4 * m + 1], h[4 * m + 2], h[4 * m + 3]};
                    *reinterpret_cast<V4*>(qrow + m * 16) = v;
                }
                const T first = q == 0 ? T(1) : T(0);
                V4 vq = {h[KT - 1], first * (sig_in == 0 ? T(1) : T(0)), first * (sig_in == 1 ? T(1) : T(0)), first};
                *reinterpret_cast<V4*>(qrow + NFULL * 16) = vq;
            }
            // dL/dh_{n-1} += W_r d a_r + W_u d a_u + Wch d q   (A = transposed weights, B = dp fragments)
            V4 accb[G::NTO];
#pragma unroll
            for (int t = 0; t < G::NTO; ++t) accb[t] = V4{T(0), T(0), T(0), T(0)};
            asm volatile("" ::: "memory");
            VA af[G::NTO], afn[G::NTO];
#pragma unroll
            for (int t = 0; t < G::NTO; ++t) af[t] = bwd_frag(t, 0);
#pragma unroll
            for (int kg = 0; kg < G::KBG; ++kg) {
                if (STREAM && kg + 1 < G::KBG) {
#pragma unroll
                    for (int t = 0; t < G::NTO; ++t) afn[t] = bwd_frag(t, kg + 1);
                }
#pragma unroll
                for (int j = 0; j < VW; ++j)
#pragma unroll
                    for (int t = 0; t < G::NTO; ++t)
                        accb[t] = Frag<T>::mfma(af[t][j], dp[VW * kg + j], accb[t]);
                if (kg + 1 < G::KBG) {
#pragma unroll
                    for (int t = 0; t < G::NTO; ++t) af[t] = STREAM ? afn[t] : bwd_frag(t, kg + 1);
                }
            }
#pragma unroll
            for (int m = 0; m < NFULL; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) dh[4 * m + r] += accb[m][r];
            dh[KT - 1] += accb[NFULL][0];
        }
    }
    if (bwd_role) store_head_part<T, NOUT, KT>(reinterpret_cast<T*>(a.head_part) + (size_t)gw * NOUT * G::HEAD_ROW, G::HEAD_ROW, hg, gb, c, q);
}

// Partial sums of  P[row][:]^T Q[row][:]  over one contiguous chunk of rows per block (4 waves).
// A 16x16x4 step takes four rows (k = lane quarter lk); which sixteen columns make up an output tile is free, so the columns of
// a row are taken in CHUNKS of VW tiles (16 bytes per lane): lane li loads the VW consecutive columns  off + VW li .. + VW - 1
// of its row with one 16-byte load and element j feeds tile j of the chunk (tile row i <-> column off + VW i + j).  The last
// chunk of a row may be narrower (P: 2 tiles at f32; Q: 1 - 3), loaded with 8- and 4-byte pieces.  Wave w owns the P chunks
// w, w + WAVES, ... and every Q chunk; the next eight rows are fetched while the current eight are multiplied.
template <typename T, int NTILES, int CW = 16 / (int)sizeof(T)>
struct RowChunks {
    static constexpr int VW = CW;                          // tiles per chunk = elements per lane and load
    static constexpr int NC = (NTILES + VW - 1) / VW;
    static constexpr int WL = NTILES - VW * (NC - 1);      // tiles of the last chunk, 1 .. VW
    static constexpr int COLS = 16 * NTILES;
    static constexpr int width(int c) { return c < NC - 1 ? VW : WL; }
    static constexpr int col(int c, int i, int j) { return 16 * VW * c + width(c) * i + j; }
    // elements of a chunk of W tiles starting at p for lane li:  v[j] = p[W li + j]
    template <int W>
    static __device__ __forceinline__ void load(const T* __restrict__ p, int li, T (&v)[VW]) {
#pragma unroll
        for (int j = W; j < VW; ++j) v[j] = T(0);
        if constexpr (W == 4 || W == 2) {
            typedef T VT __attribute__((ext_vector_type(W)));
            const VT x = *reinterpret_cast<const VT*>(p + W * li);
#pragma unroll
            for (int j = 0; j < W; ++j) v[j] = x[j];
        } else {
#pragma unroll
            for (int j = 0; j < W; ++j) v[j] = p[W * li + j];
        }
    }
};

// Waves per block and P-chunk width of the product: 8 waves where four would hold more than 160 accumulator registers each;
// the widest chunk (fewest loads) among those that give the busiest wave the fewest tiles (the large shapes are MFMA-bound).
constexpr int tn_max_tiles(int nt, int cw, int waves) {
    const int nc = (nt + cw - 1) / cw;
    int worst = 0;
    for (int w = 0; w < waves; ++w) {
        int tiles = 0;
        for (int c = w; c < nc; c += waves) tiles += c < nc - 1 ? cw : nt - cw * (nc - 1);
        worst = tiles > worst ? tiles : worst;
    }
    return worst;
}
template <typename T, int PT, int QT>
struct TnGemmShape {
    static constexpr int VWMAX = 16 / (int)sizeof(T);
    static constexpr int WAVES = ((PT + 3) / 4) * QT * (int)sizeof(T) > 160 ? 8 : 4;
    static constexpr int pick() {
        int best = VWMAX;
        for (int cw = VWMAX / 2; cw >= 1; cw /= 2)
            if (tn_max_tiles(PT, cw, WAVES) < tn_max_tiles(PT, best, WAVES)) best = cw;
        return best;
    }
    static constexpr int CW = pick();
    using CP = RowChunks<T, PT, CW>;
    using CQ = RowChunks<T, QT>;
};

// LAST: what this wave's chunk slot ILAST = (NC - 1) / WAVES holds: 0 nothing, 1 the row's narrow last chunk, 2 a full chunk.
template <typename T, int PT, int QT, int WAVES, int LAST>
__device__ __forceinline__ void tn_gemm_body(const T* __restrict__ P, const T* __restrict__ Q, int64_t r0, int64_t r1,
                                             T* __restrict__ part, int wave, int li, int lk, int q_stride) {
    using V4 = typename Frag<T>::V4;
    using CP = typename TnGemmShape<T, PT, QT>::CP;
    using CQ = typename TnGemmShape<T, PT, QT>::CQ;
    constexpr int VW = CP::VW, VQ = CQ::VW;
    constexpr int ILAST = (CP::NC - 1) / WAVES;
    constexpr int MC = ILAST + (LAST != 0 ? 1 : 0);        // P chunks of this wave
    constexpr int NS = MC * VW * QT * (int)sizeof(T) > 160 ? 1 : 2;    // 4-row steps per iteration (one when the accumulators are many)
    if constexpr (MC > 0) {
        auto pw = [](int i) { return i == ILAST && LAST == 1 ? CP::WL : VW; };
        V4 acc[MC][VW][QT];
#pragma unroll
        for (int i = 0; i < MC; ++i)
#pragma unroll
            for (int jp = 0; jp < VW; ++jp)
#pragma unroll
                for (int t = 0; t < QT; ++t) acc[i][jp][t] = V4{T(0), T(0), T(0), T(0)};
        T pa[NS][MC][VW], qb[NS][CQ::NC][VQ];
        // rows past r1 are read from row r0 instead (always in bounds) and zeroed when they are multiplied
        auto fetch = [&](int64_t r, T (&pv)[NS][MC][VW], T (&qv)[NS][CQ::NC][VQ], bool (&okv)[NS]) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t row = r + 4 * s + lk;
                const bool ok = row < r1;
                okv[s] = ok;
                const int64_t rc = ok ? row : r0;
                const T* prow = P + rc * CP::COLS + 16 * VW * wave;
#pragma unroll
                for (int i = 0; i < MC; ++i) {
                    if (i == ILAST && LAST == 1) CP::template load<CP::WL>(prow + 16 * VW * WAVES * i, li, pv[s][i]);
                    else CP::template load<VW>(prow + 16 * VW * WAVES * i, li, pv[s][i]);
                }
                const T* qrow = Q + rc * q_stride;          // (Q points at this launch's first column: tn_gemm.h slices wide products)
#pragma unroll
                for (int c = 0; c < CQ::NC; ++c) {
                    if (c == CQ::NC - 1) CQ::template load<CQ::WL>(qrow + 16 * VQ * c, li, qv[s][c]);
                    else CQ::template load<VQ>(qrow + 16 * VQ * c, li, qv[s][c]);
                }
            }
        };
        auto multiply = [&](const T (&pv)[NS][MC][VW], const T (&qv)[NS][CQ::NC][VQ], const bool (&okv)[NS]) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int i = 0; i < MC; ++i)
#pragma unroll
                    for (int jp = 0; jp < VW; ++jp) {
                        if (jp >= pw(i)) continue;
                        const T a = okv[s] ? pv[s][i][jp] : T(0);
#pragma unroll
                        for (int t = 0; t < QT; ++t)
                            acc[i][jp][t] = Frag<T>::mfma(a, qv[s][t / VQ][t % VQ], acc[i][jp][t]);
                    }
        };
        // two register sets: the loads of one are in flight while the other is multiplied (a set past r1 is all zeros)
        T pb[NS][MC][VW], qc[NS][CQ::NC][VQ];
        bool oka[NS], okb[NS];
        fetch(r0, pa, qb, oka);
        for (int64_t r = r0; r < r1; r += 8 * NS) {
            fetch(r + 4 * NS, pb, qc, okb);
            __builtin_amdgcn_sched_barrier(0);
            multiply(pa, qb, oka);
            __builtin_amdgcn_sched_barrier(0);
            fetch(r + 8 * NS, pa, qb, oka);
            __builtin_amdgcn_sched_barrier(0);
            multiply(pb, qc, okb);
            __builtin_amdgcn_sched_barrier(0);
        }
        // this block's partial sums, one 16-byte fragment per (tile, lane): part[block][P tile][Q tile][64]
        V4* out = reinterpret_cast<V4*>(part) + (size_t)blockIdx.x * PT * QT * 64 + (threadIdx.x & 63);
#pragma unroll
        for (int i = 0; i < MC; ++i)
#pragma unroll
            for (int jp = 0; jp < VW; ++jp) {
                if (jp >= pw(i)) continue;
                const int ptile = VW * (wave + WAVES * i) + jp;
#pragma unroll
                for (int t = 0; t < QT; ++t) out[(size_t)(ptile * QT + t) * 64] = acc[i][jp][t];
            }
    }
}

template <typename T, int PT, int QT>
__global__ void __launch_bounds__((TnGemmShape<T, PT, QT>::WAVES * 64)) tn_gemm_kernel(const T* __restrict__ P, const T* __restrict__ Q, int64_t R,
                                                                                    int64_t rows_per_block, T* __restrict__ part, int q_stride) {
    using CP = typename TnGemmShape<T, PT, QT>::CP;
    constexpr int WAVES = TnGemmShape<T, PT, QT>::WAVES;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;      // the host launches ceil(R / rows_per_block) blocks: r0 < R
    const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    constexpr int WLAST = (CP::NC - 1) % WAVES;            // the wave that owns the row's last chunk
    if (wave < WLAST) tn_gemm_body<T, PT, QT, WAVES, 2>(P, Q, r0, r1, part, wave, li, lk, q_stride);
    else if (wave == WLAST) tn_gemm_body<T, PT, QT, WAVES, CP::WL == CP::VW ? 2 : 1>(P, Q, r0, r1, part, wave, li, lk, q_stride);
    else tn_gemm_body<T, PT, QT, WAVES, 0>(P, Q, r0, r1, part, wave, li, lk, q_stride);
}

// dW[P column][Q column] += sum over the blocks' partial fragments, in a fixed order (no atomics: the weight gradients are
// bit-reproducible).  One workgroup of 16 waves per output tile: wave w adds the blocks w, w + 16, ..., wave 0 the sixteen sums.
template <typename T, int PT, int QT>
__global__ void __launch_bounds__(1024) tn_reduce_kernel(const T* __restrict__ part, int nblocks, T* __restrict__ dW, int dw_stride) {
    using V4 = typename Frag<T>::V4;
    using CP = typename TnGemmShape<T, PT, QT>::CP;
    using CQ = typename TnGemmShape<T, PT, QT>::CQ;
    constexpr int VW = CP::VW, VQ = CQ::VW;
    __shared__ V4 sums[16][64];
    const int tile = blockIdx.x, ptile = tile / QT, t = tile % QT;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const V4* src = reinterpret_cast<const V4*>(part) + (size_t)tile * 64 + lane;
    V4 v = V4{T(0), T(0), T(0), T(0)};
    for (int b = w; b < nblocks; b += 16) v += src[(size_t)b * PT * QT * 64];
    sums[w][lane] = v;
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) v += sums[k][lane];
    const int li = lane & 15, lk = lane >> 4;
    const int c = ptile / VW, jp = ptile % VW, cq = t / VQ;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int fr = sizeof(T) == 4 ? 4 * lk + rr : lk + 4 * rr;      // C/D fragment row of (lane quarter lk, register rr)
        dW[(size_t)CP::col(c, fr, jp) * dw_stride + CQ::col(cq, li, t % VQ)] += v[rr];      // (dW points at this launch's first column)
    }
}

}  // namespace rnnwf
