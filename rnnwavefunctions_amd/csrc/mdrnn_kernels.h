// mdrnn_kernels.h - 2D vanilla RNN wave function on the zig-zag path, float64
// (2DTFIM_2DRNN/MDRNNcell.py:51-66, 2DTFIM_2DRNN/RNNwavefunction.py:35-200).
//
// One step:  h' = elu(x_h Uh + h_h Wh + x_v Uv + h_v Wv + b)  as  D^T = [Wh^T | Wv^T] [h_h ; h_v]  on the
// f64 16x16x4 MFMA (C/D row = q + 4 r, so the fragment order is the natural unit order and, as for the GRU,
// the output fragment is directly the next step's B operand).  One-hot inputs fold into the accumulator
// initialisation.  Hidden states of all sites live in HBM in fragment order:
//   hs [N][nsb][KT][64] f64  state after visit position p (base pass: out; flip pass: in)
// because every site needs its vertical neighbour from the previous row.  A flip chain re-evaluates
// positions i+1..N-1; states it produces itself go to a private ring of 2*Nx positions per wave.
#pragma once
#include "device.h"

namespace rnnwf {

template <int NFULL_>
struct MdLayout {
    static constexpr int NFULL = NFULL_;
    static constexpr int KT = 4 * NFULL + 1;          // k-steps per hidden vector
    static constexpr int NT = NFULL + 1;              // 16-row output tiles (last: units 16 NFULL + q in reg 0)
    static constexpr size_t OFF_A = 0;                                       // [NT][KT][64] double2 (k-steps 2g, 2g+1 of [h_h ; h_v])
    static constexpr size_t OFF_BH = OFF_A + (size_t)NT * KT * 64 * 16;      // [3][NT][4][4] f64 : b + Uh[x_h]
    static constexpr size_t SZ_B = ((size_t)NT * 16 + 4) * 8;
    static constexpr size_t OFF_BV = OFF_BH + 3 * SZ_B;                      // [3][NT][4][4] f64 : Uv[x_v]
    static constexpr size_t OFF_WD = OFF_BV + 3 * SZ_B;                      // [KT][4][2] f64
    static constexpr size_t OFF_BD = OFF_WD + (size_t)KT * 4 * 2 * 8;        // [2] f64
    static constexpr size_t BYTES = ((OFF_BD + 16 + 15) / 16) * 16;
};

struct MdArgs {
    const void* wimg;
    int32_t N, Nx;
    int64_t ns, nsb;
    uint32_t* bits;                // spins in visit order
    double* hs;                    // [N][nsb][KT][64]
    double* ring;                  // flip pass: [total waves][2 Nx][KT][64] private states
    double* lpq;                   // [N+1][ns]
    double* out_lp;                // [ns]
    const int32_t* vert_pos;       // [N] visit position of the vertical neighbour, -1 at the first row
    const int32_t* row_first;      // [N] 1 if the site is the first visited of its row (no horizontal neighbour)
    const int32_t* row_of_pos;     // [N] lpq row of a flip at visit position p (= nx*Ny + ny + 1)
    uint64_t seed, step;
    int64_t sample_offset;
    int32_t sampling;
    int64_t ntiles;
};

template <int NFULL>
struct MdCore {
    using L = MdLayout<NFULL>;
    using F = Frag<double>;
    using V4 = F::V4;
    typedef double V2 __attribute__((ext_vector_type(2)));
    static constexpr int KT = L::KT, NT = L::NT;

    static __device__ __forceinline__ void stage(char* lds, const void* wimg) {
        const uint4* src = reinterpret_cast<const uint4*>(wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < (int)(L::BYTES / 16); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }

    // hk[0..KT) = h_h fragment, hk[KT..2KT) = h_v fragment; result in out[KT]
    static __device__ __forceinline__ void step(const char* lds, int sig_h, int sig_v, const double (&hk)[2 * KT],
                                                double (&out)[KT], int lane) {
        const int q = lane >> 4;
        asm volatile("" ::: "memory");   // keep the weight fragments in LDS, not in registers (see gru_core.h)
        V4 acc[NT];
        {
            const char* bh = lds + L::OFF_BH + (size_t)(sig_h + 1) * L::SZ_B + (size_t)q * 32;
            const char* bv = lds + L::OFF_BV + (size_t)(sig_v + 1) * L::SZ_B + (size_t)q * 32;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = *reinterpret_cast<const V4*>(bh + t * 128) + *reinterpret_cast<const V4*>(bv + t * 128);
        }
        const V2* av = reinterpret_cast<const V2*>(lds + L::OFF_A) + lane;
#pragma unroll
        for (int g = 0; g < KT; ++g) {
            V2 a[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) a[t] = av[(t * KT + g) * 64];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = F::mfma(a[t][j], hk[2 * g + j], acc[t]);
        }
#pragma unroll
        for (int m = 0; m < NFULL; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double x = acc[m][r];
                out[4 * m + r] = x > 0.0 ? x : expm1(x);          // tf.nn.elu
            }
        {
            const double x = acc[NT - 1][0];
            out[KT - 1] = x > 0.0 ? x : expm1(x);
        }
    }

    static __device__ __forceinline__ void head(const char* lds, const double (&h)[KT], int lane, double& p0, double& p1) {
        const int q = lane >> 4;
        asm volatile("" ::: "memory");
        const double* wd = reinterpret_cast<const double*>(lds + L::OFF_WD) + q * 2;
        double z0 = 0.0, z1 = 0.0;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            z0 += h[kt] * wd[kt * 8];
            z1 += h[kt] * wd[kt * 8 + 1];
        }
        const double* bd = reinterpret_cast<const double*>(lds + L::OFF_BD);
        z0 += __shfl_xor(z0, 16); z0 += __shfl_xor(z0, 32); z0 += bd[0];
        z1 += __shfl_xor(z1, 16); z1 += __shfl_xor(z1, 32); z1 += bd[1];
        const double m = z0 > z1 ? z0 : z1;
        const double e0 = exp(z0 - m), e1 = exp(z1 - m);
        p0 = e0 / (e0 + e1);
        p1 = e1 / (e0 + e1);
    }
};

__device__ __forceinline__ int md_spin(const uint32_t* bits, int64_t ns, int64_t s, int p) {
    return (int)((bits[(int64_t)(p >> 5) * ns + s] >> (p & 31)) & 1);
}

template <int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) mdrnn_base_kernel(MdArgs a) {
    using C = MdCore<NFULL>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, a.wimg);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int W = (N + 31) / 32;
    for (int64_t sb = gw; sb < a.nsb; sb += nw) {
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        double hk[2 * KT];      // [h_h | h_v]
        double hn[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) hn[kt] = 0.0;
        uint32_t words[8];      // spins drawn / read so far (N <= 256)
#pragma unroll
        for (int w = 0; w < 8; ++w) words[w] = (!a.sampling && w < W) ? a.bits[(int64_t)w * a.ns + sc] : 0u;
        int sig_prev = -1;
        double cum = 0.0;
        for (int p = 0; p < N; ++p) {
            const int pv = a.vert_pos[p];
            const bool first = a.row_first[p] != 0;
            // horizontal neighbour = previous visit unless this site starts a row
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) hk[kt] = first ? 0.0 : hn[kt];
            const int sig_h = first ? -1 : sig_prev;
            int sig_v = -1;
            if (pv < 0) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hk[KT + kt] = 0.0;
            } else if (pv == p - 1) {           // row turn: the vertical neighbour was computed one step ago
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hk[KT + kt] = hn[kt];
                sig_v = sig_prev;
            } else {
                const double* src = a.hs + (((int64_t)pv * a.nsb + sb) * KT) * 64 + lane;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hk[KT + kt] = src[kt * 64];
                uint32_t wv = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) if (w == (pv >> 5)) wv = words[w];
                sig_v = (wv >> (pv & 31)) & 1;
            }
            C::step(lds, sig_h, sig_v, hk, hn, lane);
            double p0, p1;
            C::head(lds, hn, lane, p0, p1);
            int sig;
            if (a.sampling) {
                const float u = philox_uniform(a.seed, a.step, (uint64_t)(a.sample_offset + sc), p);
                sig = ((double)u * (p0 + p1) < p0) ? 0 : 1;
#pragma unroll
                for (int w = 0; w < 8; ++w) if (w == (p >> 5)) words[w] |= (uint32_t)sig << (p & 31);
            } else {
                uint32_t wp = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) if (w == (p >> 5)) wp = words[w];
                sig = (wp >> (p & 31)) & 1;
            }
            const double lsel = log(sig ? p1 : p0);
            if (a.lpq && valid && q == 0) a.lpq[(int64_t)a.row_of_pos[p] * a.ns + s] = cum + log(sig ? p0 : p1);
            cum += lsel;
            if (a.hs) {
                double* dst = a.hs + (((int64_t)p * a.nsb + sb) * KT) * 64 + lane;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) dst[kt * 64] = hn[kt];
            }
            sig_prev = sig;
        }
        if (valid && q == 0) {
            if (a.sampling)
#pragma unroll
                for (int w = 0; w < 8; ++w) if (w < W) a.bits[(int64_t)w * a.ns + s] = words[w];
            if (a.lpq) a.lpq[s] = cum;
            if (a.out_lp) a.out_lp[s] = cum;
        }
    }
}

template <int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) mdrnn_flip_kernel(MdArgs a) {
    using C = MdCore<NFULL>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, a.wimg);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int W = (N + 31) / 32;
    const int R = 2 * a.Nx;                                            // ring slots (positions) per wave
    double* ring = a.ring + (int64_t)gw * R * KT * 64 + lane;
    for (int64_t tile = gw; tile < a.ntiles; tile += nw) {
        const int i = (int)(tile / a.nsb);
        const int64_t sb = tile - (int64_t)i * a.nsb;
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        uint32_t words[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) words[w] = w < W ? a.bits[(int64_t)w * a.ns + sc] : 0u;
#pragma unroll
        for (int w = 0; w < 8; ++w) if (w == (i >> 5)) words[w] ^= 1u << (i & 31);    // the flipped configuration
        double hk[2 * KT], hn[KT];
        {
            const double* src = a.hs + (((int64_t)i * a.nsb + sb) * KT) * 64 + lane;  // state after position i (unchanged)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) hn[kt] = src[kt * 64];
        }
        double lp = 0.0;
        for (int p = i + 1; p < N; ++p) {
            const int pv = a.vert_pos[p];
            const bool first = a.row_first[p] != 0;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) hk[kt] = first ? 0.0 : hn[kt];
            uint32_t wq = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) if (w == ((p - 1) >> 5)) wq = words[w];
            const int sig_h = first ? -1 : (int)((wq >> ((p - 1) & 31)) & 1);
            int sig_v = -1;
            if (pv < 0) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hk[KT + kt] = 0.0;
            } else {
                if (pv == p - 1) {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) hk[KT + kt] = hn[kt];
                } else if (pv <= i) {        // produced by the base pass
                    const double* src = a.hs + (((int64_t)pv * a.nsb + sb) * KT) * 64 + lane;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) hk[KT + kt] = src[kt * 64];
                } else {                     // produced by this chain
                    const double* src = ring + (int64_t)(pv % R) * KT * 64;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) hk[KT + kt] = src[kt * 64];
                }
                uint32_t wv = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) if (w == (pv >> 5)) wv = words[w];
                sig_v = (wv >> (pv & 31)) & 1;
            }
            C::step(lds, sig_h, sig_v, hk, hn, lane);
            double p0, p1;
            C::head(lds, hn, lane, p0, p1);
            uint32_t wp = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) if (w == (p >> 5)) wp = words[w];
            lp += log(((wp >> (p & 31)) & 1) ? p1 : p0);
            {
                double* dst = ring + (int64_t)(p % R) * KT * 64;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) dst[kt * 64] = hn[kt];
            }
        }
        if (valid && q == 0) a.lpq[(int64_t)a.row_of_pos[i] * a.ns + s] += lp;
    }
}

}  // namespace rnnwf
