// ml_grad_kernels.h - back-propagation through a stacked GRU layer above the first (f32).
//
// The gradient of a stack is taken layer by layer, top first: the pass of layer l walks the sites N-1 .. 0, re-computes
// the layer's gates from the stored states (x = new state of layer l-1 at this site, h = own state of the previous
// site; the base pass stored every layer's state of every site), and back-propagates
//     dL/dh_{n-1} = u * d + [Wg_h(r) Wg_h(u) Wch] [da_r; da_u; dq]      (carried in registers, as in gru_bwd_kernel)
//     dL/dx_n     =         [Wg_x(r) Wg_x(u) Wci] [da_r; da_u; dy]      (written to HBM: the dh_in of layer l-1's pass)
// with both products on the f32 MFMA.  dL/dh_n comes from the head (top layer) or from the pass of the layer above.
// Weight gradients are P^T Q with the same P rows as the first layer and Q = [x | h_{n-1}, 1].
// One pass needs only this layer's images in LDS: forward 67 KB + two backward operands 80 KB at 50 units.
#pragma once
#include "grad_kernels.h"

namespace rnnwf {

template <int NFULL, int NOUT = 1, typename T = float>
struct UpperGradLayout {
    static constexpr int KT = 4 * NFULL + 1;
    static constexpr int NT = 3 * NFULL + 1;
    static constexpr int NTO = NFULL + 1;
    static constexpr int PCOLS = 16 * (NT + NTO);         // as GradLayout: [gate rows in image order | dy]
    static constexpr int QCOLS = 32 * NTO;                // [x | h_{n-1} with the constant 1 in a spare slot]
    static constexpr int KB = 3 * KT;
    static constexpr int VW = 16 / (int)sizeof(T);        // k-steps per 16-byte LDS vector
    static constexpr int KBG = (KB + VW - 1) / VW;
    static constexpr size_t SIDE_BYTES = (size_t)NTO * KBG * 64 * 16;    // [NTO][KBG][64] x 16 B
    static constexpr size_t BWD_BYTES = 2 * SIDE_BYTES;                  // H side (-> dh), X side (-> dx)
    static constexpr int WD_Q = ((NOUT * KT + 3) / 4) * 4;               // GruLayout<T, NFULL, NOUT>::WD_Q
    static constexpr size_t HEAD_BYTES = ((size_t)4 * WD_Q * sizeof(T) + 32 + 15) / 16 * 16;   // [4 q][KT][NOUT] + bias (OFF_WD .. end of the image)
    static constexpr int HEAD_ROW = 4 * KT + 4;                          // as GradLayout::HEAD_ROW
    // forward image + both backward operands + head rows beyond the 160 KB of LDS (53..100 units): read through L2 instead
    static constexpr bool WIDE = UpperLayout<NFULL, T>::BYTES + BWD_BYTES + HEAD_BYTES > 160 * 1024;
    static constexpr size_t LDS_BYTES = WIDE ? HEAD_BYTES : UpperLayout<NFULL, T>::BYTES + BWD_BYTES + HEAD_BYTES;
};

struct UpperGradArgs {
    const void* wup;           // forward image of this layer (UpperLayout)
    const void* wbwd;          // UpperGradLayout::BWD_BYTES
    const void* whead;         // OFF_WD section of the first layer's image (head weights), top layer only
    int32_t N, layer, hck_nl;
    int64_t ns, nsb;
    const uint32_t* bits;
    const void* hck;           // [N][nsb][NL][KT][64] T
    const double* eloc;        // [ns] f64 (positive RNN) ...
    const float2* eloc_c;      // ... or [ns] complex64 (complex RNN)
    double mean_e, mean_im, inv_norm;
    const double* mom;         // device-resident training: the step's moments on the device (grad_kernels.h: GradArgs::mom), nullptr: the fields above
    const double* wfac;        // [ns] extra factor of w_s or nullptr (parity-symmetric model: the direction's share of P_sym)
    const void* dh_in;         // [N][nsb][KT][64] T from the layer above (nullptr: top layer, head)
    void* dx_out;              // [N][nsb][KT][64] T
    void* P;
    void* Q;
    void* head_grad;           // [NOUT][HEAD_ROW] T (top layer; written by head_reduce_kernel)
    void* head_part;           // [waves of the grid][NOUT][HEAD_ROW] T (top layer)
};

// NOUT (top layer only): 1 = positive RNN head, 3 = complex RNN heads (the site terms of gru_bwd_kernel).
template <typename T, int NFULL, int WAVES, bool TOP, int NOUT = 1>
__global__ void __launch_bounds__(WAVES * 64) gru_upper_bwd_kernel(UpperGradArgs a) {
    using CU = UpperCore<NFULL, T>;
    using G = UpperGradLayout<NFULL, NOUT, T>;
    using V4 = typename CU::V4;
    using VA = typename CU::VA;
    constexpr int KT = CU::KT, NT = G::NT, VW = G::VW;
    const T* hck = reinterpret_cast<const T*>(a.hck);
    const T* dh_in = reinterpret_cast<const T*>(a.dh_in);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // WIDE (53..100 units): the layer's forward image and its two backward operands exceed LDS; both stay in global memory and are
    // read through L2 with buffer loads (forward: UpperCore::block's bounded look-ahead; backward: one k-group ahead); LDS holds
    // the head rows only
    constexpr bool WIDE = G::WIDE;
    constexpr size_t HEAD_OFF = WIDE ? 0 : CU::U::BYTES + G::BWD_BYTES;
    {
        auto copy = [&](char* dst_, const void* src_, size_t bytes) {
            const uint4* src = reinterpret_cast<const uint4*>(src_);
            uint4* dst = reinterpret_cast<uint4*>(dst_);
            for (int i = threadIdx.x; i < (int)(bytes / 16); i += blockDim.x) dst[i] = src[i];
        };
        if constexpr (!WIDE) {
            copy(lds, a.wup, CU::U::BYTES);
            copy(lds + CU::U::BYTES, a.wbwd, G::BWD_BYTES);
        }
        if (TOP) copy(lds + HEAD_OFF, a.whead, G::HEAD_BYTES);
        __syncthreads();
    }
    const char* fwd = WIDE ? reinterpret_cast<const char*>(a.wup) : lds;
    const char* lbh = lds + CU::U::BYTES;
    const char* lbx = lbh + G::SIDE_BYTES;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rbwd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wbwd), 0, (int)G::BWD_BYTES, 0x00020000);
    auto bwd_frag = [&](int side, int t, int kg) -> VA {    // side 0: H operand, 1: X operand
        if constexpr (WIDE) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rbwd, (threadIdx.x & 63) * 16, side * (int)G::SIDE_BYTES + (t * G::KBG + kg) * 64 * 16, 0);
            return __builtin_bit_cast(VA, v);
        } else {
            return (reinterpret_cast<const VA*>(side ? lbx : lbh) + (threadIdx.x & 63))[(t * G::KBG + kg) * 64];
        }
    };
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const T* wd = reinterpret_cast<const T*>(lds + HEAD_OFF) + q * G::WD_Q;
    const T* bd = reinterpret_cast<const T*>(lds + HEAD_OFF) + 4 * G::WD_Q;
    T hg[NOUT][KT], gb[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        gb[o] = T(0);
#pragma unroll
        for (int k = 0; k < KT; ++k) hg[o][k] = T(0);
    }
    for (int64_t sb = gw; sb < a.nsb; sb += nw) {
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        T w = T(0), w_im = T(0);
        if (TOP && valid) {
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
        auto spin = [&](int n) { return (int)((a.bits[(int64_t)(n >> 5) * a.ns + sc] >> (n & 31)) & 1); };
        int num_up = 0;                                   // complex RNN: up spins among sites < n (U(1) mask)
        if constexpr (TOP && NOUT == 3)
            for (int m = 0; m < N; ++m) num_up += spin(m);
        T dh[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) dh[k] = T(0);
        // the inputs of site n - 1 (state of the layer below at that site, this layer's state before it) are fetched while
        // site n is worked on
        auto fetch_inputs = [&](int n, T (&xd)[KT], T (&hd)[KT]) {
            {
                const T* src = hck + ((((int64_t)n * a.nsb + sb) * a.hck_nl + a.layer - 1) * KT) * 64 + lane;
#pragma unroll
                for (int k = 0; k < KT; ++k) xd[k] = src[k * 64];
            }
            if (n > 0) {
                const T* src = hck + ((((int64_t)(n - 1) * a.nsb + sb) * a.hck_nl + a.layer) * KT) * 64 + lane;
#pragma unroll
                for (int k = 0; k < KT; ++k) hd[k] = src[k * 64];
            } else {
#pragma unroll
                for (int k = 0; k < KT; ++k) hd[k] = T(0);
            }
        };
        T xpf[KT], hpf[KT];
        fetch_inputs(N - 1, xpf, hpf);
        for (int n = N - 1; n >= 0; --n) {
            T x[KT], h[KT], hn[KT], rg[KT], ug[KT], cc[KT], qv[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) { x[k] = xpf[k]; h[k] = hpf[k]; }
            if (n > 0) fetch_inputs(n - 1, xpf, hpf);
            CU::step_keep(fwd, x, h, hn, rg, ug, cc, qv, lane, WIDE);
            T g[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) g[o] = T(0);
            if (TOP) {
                asm volatile("" ::: "memory");
                T z[NOUT];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) z[o] = T(0);
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) z[o] += hn[k] * wd[k * NOUT + o];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    z[o] += __shfl_xor(z[o], 16); z[o] += __shfl_xor(z[o], 32); z[o] += bd[o];
                }
                const int sig = spin(n);
                const T p1 = T(1) - prob0(z[0]);
                if constexpr (NOUT == 1) {
                    g[0] = w * ((T)sig - p1);                // d log p(sig) / d(z1 - z0) = sig - p1
                } else {                                         // the site terms of gru_bwd_kernel<.., 3>
                    num_up -= sig;
                    bool both = true;
                    if (2 * n >= N) {
                        const int base = N / 2 - 1;
                        both = (base - (n - num_up) >= 0) && (base - num_up >= 0);
                    }
                    g[0] = both ? T(0.5) * w * ((T)sig - p1) : T(0);
                    const T zs = sig ? z[2] : z[1];
                    const T den = T(1) + (zs < T(0) ? -zs : zs);
                    const T gp = w_im * T(3.14159265358979323846) / (den * den);
                    g[1] = sig ? T(0) : gp;
                    g[2] = sig ? gp : T(0);
                }
#pragma unroll
                for (int o = 0; o < NOUT; ++o) gb[o] += g[o];
            }
            T dpH[VW * G::KBG], dpX[VW * G::KBG];
            T dy[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                T d = dh[k];
                if (TOP) {
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) {
                        hg[o][k] += g[o] * hn[k];
                        d += g[o] * wd[k * NOUT + o];
                    }
                } else {
                    d += dh_in[(((int64_t)n * a.nsb + sb) * KT + k) * 64 + lane];
                }
                const T du = d * (h[k] - cc[k]);
                const T dc = d * (T(1) - ug[k]);
                dh[k] = d * ug[k];
                dy[k] = dc * (T(1) - cc[k] * cc[k]);
                dpH[k] = dpX[k] = dy[k] * qv[k] * rg[k] * (T(1) - rg[k]);     // d a_r
                dpH[KT + k] = dpX[KT + k] = du * ug[k] * (T(1) - ug[k]);     // d a_u
                dpH[2 * KT + k] = dy[k] * rg[k];                             // d q
                dpX[2 * KT + k] = dy[k];                                     // d y
            }
#pragma unroll
            for (int k = G::KB; k < VW * G::KBG; ++k) dpH[k] = dpX[k] = T(0);
            if (valid) {
                T* prow = reinterpret_cast<T*>(a.P) + ((int64_t)n * a.ns + s) * G::PCOLS + 4 * q;
#pragma unroll
                for (int m = 0; m < NFULL; ++m) {
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt)
                        *reinterpret_cast<V4*>(prow + (gt * NFULL + m) * 16) =
                            V4{dpH[gt * KT + 4 * m], dpH[gt * KT + 4 * m + 1], dpH[gt * KT + 4 * m + 2], dpH[gt * KT + 4 * m + 3]};
                    *reinterpret_cast<V4*>(prow + (NT + m) * 16) = V4{dy[4 * m], dy[4 * m + 1], dy[4 * m + 2], dy[4 * m + 3]};
                }
                *reinterpret_cast<V4*>(prow + (NT - 1) * 16) = V4{dpH[KT - 1], dpH[2 * KT - 1], dpH[3 * KT - 1], T(0)};
                *reinterpret_cast<V4*>(prow + (NT + NFULL) * 16) = V4{dy[KT - 1], T(0), T(0), T(0)};
                T* qrow = reinterpret_cast<T*>(a.Q) + ((int64_t)n * a.ns + s) * G::QCOLS + 4 * q;
#pragma unroll
                for (int m = 0; m < NFULL; ++m) {
                    *reinterpret_cast<V4*>(qrow + m * 16) = V4{x[4 * m], x[4 * m + 1], x[4 * m + 2], x[4 * m + 3]};
                    *reinterpret_cast<V4*>(qrow + (G::NTO + m) * 16) = V4{h[4 * m], h[4 * m + 1], h[4 * m + 2], h[4 * m + 3]};
                }
                *reinterpret_cast<V4*>(qrow + NFULL * 16) = V4{x[KT - 1], T(0), T(0), T(0)};
                *reinterpret_cast<V4*>(qrow + (G::NTO + NFULL) * 16) = V4{h[KT - 1], q == 0 ? T(1) : T(0), T(0), T(0)};
            }
            V4 accH[G::NTO], accX[G::NTO];
#pragma unroll
            for (int t = 0; t < G::NTO; ++t) accH[t] = accX[t] = V4{T(0), T(0), T(0), T(0)};
            asm volatile("" ::: "memory");
            VA afh[G::NTO], afx[G::NTO], nfh[G::NTO], nfx[G::NTO];
#pragma unroll
            for (int t = 0; t < G::NTO; ++t) {
                afh[t] = bwd_frag(0, t, 0);
                afx[t] = bwd_frag(1, t, 0);
            }
#pragma unroll
            for (int kg = 0; kg < G::KBG; ++kg) {
                if (WIDE && kg + 1 < G::KBG) {
#pragma unroll
                    for (int t = 0; t < G::NTO; ++t) {
                        nfh[t] = bwd_frag(0, t, kg + 1);
                        nfx[t] = bwd_frag(1, t, kg + 1);
                    }
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int j = 0; j < VW; ++j)
#pragma unroll
                    for (int t = 0; t < G::NTO; ++t) {
                        accH[t] = Frag<T>::mfma(afh[t][j], dpH[VW * kg + j], accH[t]);
                        accX[t] = Frag<T>::mfma(afx[t][j], dpX[VW * kg + j], accX[t]);
                    }
                if (kg + 1 < G::KBG) {
#pragma unroll
                    for (int t = 0; t < G::NTO; ++t) {
                        afh[t] = WIDE ? nfh[t] : bwd_frag(0, t, kg + 1);
                        afx[t] = WIDE ? nfx[t] : bwd_frag(1, t, kg + 1);
                    }
                }
            }
            T* dxo = reinterpret_cast<T*>(a.dx_out) + (((int64_t)n * a.nsb + sb) * KT) * 64 + lane;
#pragma unroll
            for (int m = 0; m < NFULL; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dh[4 * m + r] += accH[m][r];
                    dxo[(4 * m + r) * 64] = accX[m][r];
                }
            dh[KT - 1] += accH[NFULL][0];
            dxo[(KT - 1) * 64] = accX[NFULL][0];
        }
    }
    if (TOP) store_head_part<T, NOUT, KT>(reinterpret_cast<T*>(a.head_part) + (size_t)gw * NOUT * G::HEAD_ROW, G::HEAD_ROW, hg, gb, c, q);
}

}  // namespace rnnwf
