// mdrnn_grad_kernels.h - back-propagation of the VMC cost through the 2D MDRNN (float64)
// (cost: 2DTFIM_2DRNN/Training2DRNN_2DTFIM.py:163; cell: MDRNNcell.py:51-66).
//
// The forward pass left the state of every visit position in HBM (hs), and  h = elu(pre)  gives
// d h / d pre = (h > 0 ? 1 : h + 1)  from the state itself, so the backward sweep needs no forward recompute:
//   dpre[p] = (dh from the horizontal successor + dh from the vertical successor + head) * elu'(h[p])
//   [dh_h ; dh_v] = [Wh ; Wv] dpre       one MFMA product, D^T = A_bwd dpre^T, 2 NT output tiles
// The horizontal contribution is carried in registers (positions are walked in reverse visit order; at a row
// turn the carried vector is the vertical one).  Vertical contributions wait in a per-wave ring of 2 Nx
// positions in HBM, written and read by the same lane.  Weight gradients are  P^T Q  (tn_gemm_kernel) with
//   P [N*ns][16 NT]  = dpre,   Q [N*ns][32 NT] = [h_h, onehot(x_h), 1 | h_v, onehot(x_v), 0].
#pragma once
#include "grad_kernels.h"
#include "mdrnn_kernels.h"

namespace rnnwf {

template <int NFULL_>
struct MdGradLayout {
    static constexpr int NFULL = NFULL_;
    static constexpr int KT = 4 * NFULL + 1, NT = NFULL + 1;
    static constexpr int NTO = 2 * NT;                     // output tiles: dh_h then dh_v
    static constexpr int KBG = (KT + 1) / 2;               // pairs of k-steps (b128 LDS reads)
    static constexpr size_t OFF_A = 0;                                        // [NTO][KBG][64] double2
    static constexpr size_t OFF_WD = OFF_A + (size_t)NTO * KBG * 64 * 16;     // [KT][4][2] f64
    static constexpr size_t OFF_BD = OFF_WD + (size_t)KT * 4 * 2 * 8;         // [2] f64
    static constexpr size_t BYTES = ((OFF_BD + 16 + 15) / 16) * 16;
    static constexpr int PCOLS = 16 * NT, QCOLS = 32 * NT;
    static constexpr int HEAD_ROW = 4 * KT + 4;            // slots 4 kt + q: kernel column; slot 4 KT: bias
};

struct MdGradArgs {
    const void* wbwd;
    int32_t N, Nx;
    int64_t ns, nsb;
    const uint32_t* bits;          // spins in visit order
    const double* hs;              // [N][nsb][KP][64] double2 states of the forward pass (mdrnn_kernels.h)
    double* ring;                  // [total waves][2 Nx][KT][64]
    const double* eloc;            // [ns]
    double mean_e, inv_norm;
    const double* mom;             // device-resident training: the step's moments on the device (grad_kernels.h: GradArgs::mom), nullptr: the fields above
    double* P;
    double* Q;
    double* head_grad;             // [2][HEAD_ROW], zeroed before the launch (written by head_reduce_kernel)
    double* head_part;             // [waves of the grid][2][HEAD_ROW]: every wave's head-row sums
    const int32_t* vert_pos;
    const int32_t* row_first;
};

template <int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) mdrnn_bwd_kernel(MdGradArgs a) {
    using G = MdGradLayout<NFULL>;
    using C = MdCore<NFULL>;
    using F = Frag<double>;
    using V4 = F::V4;
    typedef double V2 __attribute__((ext_vector_type(2)));
    constexpr int KT = G::KT, NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.wbwd);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < (int)(G::BYTES / 16); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int R = 2 * a.Nx;
    double* ring = a.ring + (int64_t)gw * R * KT * 64 + lane;
    const double* wd = reinterpret_cast<const double*>(lds + G::OFF_WD) + q * 2;
    const double* bd = reinterpret_cast<const double*>(lds + G::OFF_BD);
    double hg[2][KT], gb[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < KT; ++k) hg[0][k] = hg[1][k] = 0.0;
    for (int64_t sb = gw; sb < a.nsb; sb += nw) {
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        const double mean_e = a.mom ? a.mom[0] / a.mom[2] : a.mean_e, inv_norm = a.mom ? a.inv_norm / a.mom[2] : a.inv_norm;
        const double w = valid ? (a.eloc[sc] - mean_e) * inv_norm : 0.0;
        double carry[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) carry[k] = 0.0;
        for (int p = N - 1; p >= 0; --p) {
            const int pv = a.vert_pos[p];
            const bool first = a.row_first[p] != 0;
            double hn[KT];
            C::load_state(a.hs + (((int64_t)p * a.nsb + sb) * C::KP) * 128 + 2 * lane, hn);
            asm volatile("" ::: "memory");
            double z0 = 0.0, z1 = 0.0;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                z0 += hn[k] * wd[k * 8];
                z1 += hn[k] * wd[k * 8 + 1];
            }
            z0 += __shfl_xor(z0, 16); z0 += __shfl_xor(z0, 32); z0 += bd[0];
            z1 += __shfl_xor(z1, 16); z1 += __shfl_xor(z1, 32); z1 += bd[1];
            const double zm = z0 > z1 ? z0 : z1;
            const double e0 = exp(z0 - zm), e1 = exp(z1 - zm);
            const double p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
            const int sig = md_spin(a.bits, a.ns, sc, p);
            const double g0 = w * ((sig == 0 ? 1.0 : 0.0) - p0);
            const double g1 = w * ((sig == 1 ? 1.0 : 0.0) - p1);
            gb[0] += g0;
            gb[1] += g1;
            // the vertical successor is the next visit at a row turn (then its term is the carried one)
            const bool from_ring = p < N - a.Nx && a.row_first[p + 1] == 0;
            double dp[2 * G::KBG];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                double d = carry[k] + g0 * wd[k * 8] + g1 * wd[k * 8 + 1];
                if (from_ring) d += ring[((int64_t)(p % R) * KT + k) * 64];
                hg[0][k] += g0 * hn[k];
                hg[1][k] += g1 * hn[k];
                dp[k] = d * (hn[k] > 0.0 ? 1.0 : hn[k] + 1.0);
            }
#pragma unroll
            for (int k = KT; k < 2 * G::KBG; ++k) dp[k] = 0.0;
            if (valid) {
                double hh[KT], hv[KT];
                int sig_h = -1, sig_v = -1;
                if (!first) {
                    C::load_state(a.hs + (((int64_t)(p - 1) * a.nsb + sb) * C::KP) * 128 + 2 * lane, hh);
                    sig_h = md_spin(a.bits, a.ns, sc, p - 1);
                } else {
#pragma unroll
                    for (int k = 0; k < KT; ++k) hh[k] = 0.0;
                }
                if (pv >= 0) {
                    C::load_state(a.hs + (((int64_t)pv * a.nsb + sb) * C::KP) * 128 + 2 * lane, hv);
                    sig_v = md_spin(a.bits, a.ns, sc, pv);
                } else {
#pragma unroll
                    for (int k = 0; k < KT; ++k) hv[k] = 0.0;
                }
                double* prow = a.P + ((int64_t)p * a.ns + s) * G::PCOLS + 4 * q;
                double* qrow = a.Q + ((int64_t)p * a.ns + s) * G::QCOLS + 4 * q;
                const double one = q == 0 ? 1.0 : 0.0;
#pragma unroll
                for (int m = 0; m < NFULL; ++m) {
                    *reinterpret_cast<V4*>(prow + m * 16) = V4{dp[4 * m], dp[4 * m + 1], dp[4 * m + 2], dp[4 * m + 3]};
                    *reinterpret_cast<V4*>(qrow + m * 16) = V4{hh[4 * m], hh[4 * m + 1], hh[4 * m + 2], hh[4 * m + 3]};
                    *reinterpret_cast<V4*>(qrow + 16 * NT + m * 16) = V4{hv[4 * m], hv[4 * m + 1], hv[4 * m + 2], hv[4 * m + 3]};
                }
                *reinterpret_cast<V4*>(prow + NFULL * 16) = V4{dp[KT - 1], 0.0, 0.0, 0.0};
                *reinterpret_cast<V4*>(qrow + NFULL * 16) =
                    V4{hh[KT - 1], one * (sig_h == 0 ? 1.0 : 0.0), one * (sig_h == 1 ? 1.0 : 0.0), one};
                *reinterpret_cast<V4*>(qrow + 16 * NT + NFULL * 16) =
                    V4{hv[KT - 1], one * (sig_v == 0 ? 1.0 : 0.0), one * (sig_v == 1 ? 1.0 : 0.0), 0.0};
            }
            V4 acc[G::NTO];
#pragma unroll
            for (int t = 0; t < G::NTO; ++t) acc[t] = V4{0.0, 0.0, 0.0, 0.0};
            asm volatile("" ::: "memory");
            const V2* ab = reinterpret_cast<const V2*>(lds + G::OFF_A) + lane;
#pragma unroll
            for (int kg = 0; kg < G::KBG; ++kg) {
                V2 af[G::NTO];
#pragma unroll
                for (int t = 0; t < G::NTO; ++t) af[t] = ab[(t * G::KBG + kg) * 64];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < G::NTO; ++t) acc[t] = F::mfma(af[t][j], dp[2 * kg + j], acc[t]);
            }
            if (pv >= 0 && pv != p - 1) {
                double* dst = ring + (int64_t)(pv % R) * KT * 64;
#pragma unroll
                for (int m = 0; m < NFULL; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(4 * m + r) * 64] = acc[NT + m][r];
                dst[(KT - 1) * 64] = acc[NT + NFULL][0];
            }
            const bool turn = first && pv >= 0;        // p-1 is this site's vertical neighbour
#pragma unroll
            for (int m = 0; m < NFULL; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) carry[4 * m + r] = first ? (turn ? acc[NT + m][r] : 0.0) : acc[m][r];
            carry[KT - 1] = first ? (turn ? acc[NT + NFULL][0] : 0.0) : acc[NFULL][0];
        }
    }
    store_head_part<double, 2, KT>(a.head_part + (size_t)gw * 2 * G::HEAD_ROW, G::HEAD_ROW, hg, gb, c, q);
}

}  // namespace rnnwf
