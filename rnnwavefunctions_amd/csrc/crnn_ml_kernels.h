// crnn_ml_kernels.h - complex RNN with NL > 1 stacked GRU layers (the reference's DEFAULT: units=[10,10],
// J1J2/ComplexRNNwavefunction.py:16; MultiRNNCell at :40; units=[num_units]*num_layers at J1J2/TrainingRNN_J1J2.py:148).
//
// Same passes as crnn_kernels.h (masked sampling / teacher-forced log-amplitude with swap bases and checkpoints; swap
// pass over the compacted items), with the layer stack of ml_kernels.h: every layer's state in registers in B-fragment
// order, the new state of layer l is the X operand of layer l + 1 (UpperCore), the three head rows read the top layer.
// f32-input MFMA, 16 chains per wave.  LDS image: [GruLayout<float, NFULL, 3> | UpperLayout<NFULL> x (NL - 1)].
//   hck [N][nsb][NL][KT][64] f32   states of all layers after site n (all N sites: ml_grad_kernels.h needs the last one)
#pragma once
#include "crnn_kernels.h"

namespace rnnwf {

template <int NFULL, int NL>
struct CrnnMlCore {
    using C0 = GruCore<float, NFULL, 3>;
    using CU = UpperCore<NFULL>;
    static constexpr int KT = C0::KT;
    static constexpr int SPILL = MlSpill<NFULL, NL, float, 3>::value;        // layout.h: top layer read through L2
    static constexpr size_t BYTES = C0::L::BYTES + (size_t)(NL - 1 - SPILL) * CU::U::BYTES;

    static __device__ __forceinline__ void stage(char* lds, const void* wimg) {
        const uint4* src = reinterpret_cast<const uint4*>(wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < (int)(BYTES / 16); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    // all layers for one site; z = (amplitude logit difference, phase logit 0, phase logit 1) of the top layer
    static __device__ __forceinline__ void step(const char* lds, const void* wimg, int sig_in, float (&h)[NL][KT], int lane, float (&z)[3]) {
        C0::step(lds, sig_in, h[0], lane);
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            const size_t off = C0::L::BYTES + (size_t)(l - 1) * CU::U::BYTES;
            if (l < NL - SPILL) CU::step(lds + off, h[l - 1], h[l], lane);
            else CU::step(reinterpret_cast<const char*>(wimg) + off, h[l - 1], h[l], lane, NFULL >= 4);      // wide layers: bounded look-ahead
        }
        C0::head(lds, h[NL - 1], lane, z);
    }
};

template <int NFULL, int NL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) crnn_ml_base_kernel(CrnnArgs a) {
    using M = CrnnMlCore<NFULL, NL>;
    constexpr int KT = M::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    M::stage(lds, a.wimg);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    // (launched with WAVES waves or fewer: a batch of fewer 16-chain blocks than the chip has SIMDs spreads over all CUs, one wave
    //  per SIMD, instead of filling a third of them with eight - the pass is N dependent steps of MFMA latency per wave)
    const int64_t wpb = blockDim.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * wpb;
    const int N = a.N;
    for (int64_t sb = gw; sb < a.nsb; sb += nw) {
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        float h[NL][KT];
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) h[l][kt] = 0.0f;
        int sig_in = -1, num_up = 0;
        uint32_t word = 0;
        double re = 0.0, im = 0.0;
        for (int n = 0; n < N; ++n) {
            if (!a.sampling && (n & 31) == 0) word = a.bits[(int64_t)(n >> 5) * a.ns + sc];
            float z[3];
            M::step(lds, a.wimg, sig_in, h, lane, z);
            float la0, la1, w0, ph0, ph1;
            crnn_site(z, n, N, num_up, la0, la1, w0, ph0, ph1);
            int sig;
            if (a.sampling) {
                const float u = philox_uniform(a.seed, a.step, (uint64_t)(a.sample_offset + sc), n);
                sig = (u < w0) ? 0 : 1;
                word |= (uint32_t)sig << (n & 31);
                if (((n & 31) == 31 || n == N - 1) && valid && q == 0) a.bits[(int64_t)(n >> 5) * a.ns + s] = word;
                if ((n & 31) == 31) word = 0;
            } else {
                sig = (word >> (n & 31)) & 1;
            }
            if (a.cb && valid && q == 0)
                a.cb[(int64_t)n * a.ns + s] = make_double2(re + (double)(sig ? la0 : la1), im + (double)(sig ? ph0 : ph1));
            re += (double)(sig ? la1 : la0);
            im += (double)(sig ? ph1 : ph0);
            if (a.hck) {
                float* dst = reinterpret_cast<float*>(a.hck) + (((int64_t)n * a.nsb + sb) * NL * KT) * 64 + lane;
#pragma unroll
                for (int l = 0; l < NL; ++l)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) dst[(l * KT + kt) * 64] = h[l][kt];
            }
            num_up += sig;
            sig_in = sig;
        }
        if (valid && q == 0) {
            if (a.tot) a.tot[s] = make_double2(re, im);
            if (a.out_amp) a.out_amp[s] = make_float2((float)re, (float)im);
            if (a.out_logp) a.out_logp[s] = 2.0 * re;
        }
    }
}

template <int NFULL, int NL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) crnn_ml_swap_kernel(CrnnArgs a) {
    using M = CrnnMlCore<NFULL, NL>;
    constexpr int KT = M::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    M::stage(lds, a.wimg);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int64_t ntiles = a.tile_start[N];
    for (int64_t tile = gw; tile < ntiles; tile += nw) {
        int lo = 0;
        {
            int l = 0, r = N;
            while (r - l > 1) {
                const int mid = (l + r) >> 1;
                if (a.tile_start[mid] <= tile) l = mid; else r = mid;
            }
            lo = l;
        }
        const int k = (int)(tile - a.tile_start[lo]) * kChains + c;
        const bool valid = k < a.cnt[lo];
        const SwapItem it = a.items[(int64_t)lo * a.cap + (valid ? k : 0)];
        const int64_t s = it.s;
        float h[NL][KT];
        {
            const float* src = reinterpret_cast<const float*>(a.hck) +
                               (((int64_t)lo * a.nsb + (s >> 4)) * NL * KT) * 64 + (q << 4) + (s & 15);
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) h[l][kt] = src[(l * KT + kt) * 64];
        }
        int num_up = 0;
        for (int w = 0; w < (lo >> 5); ++w) num_up += __popc(a.bits[(int64_t)w * a.ns + s]);
        uint32_t word = a.bits[(int64_t)(lo >> 5) * a.ns + s];
        num_up += __popc(word & ((1u << (lo & 31)) - 1u));
        int sig_in = 1 - (int)((word >> (lo & 31)) & 1);
        num_up += sig_in;
        double re = 0.0, im = 0.0;
        for (int n = lo + 1; n < N; ++n) {
            if ((n & 31) == 0) word = a.bits[(int64_t)(n >> 5) * a.ns + s];
            float z[3];
            M::step(lds, a.wimg, sig_in, h, lane, z);
            float la0, la1, w0, ph0, ph1;
            crnn_site(z, n, N, num_up, la0, la1, w0, ph0, ph1);
            const int sig = (int)((word >> (n & 31)) & 1) ^ (n == it.hi ? 1 : 0);
            re += (double)(sig ? la1 : la0);
            im += (double)(sig ? ph1 : ph0);
            num_up += sig;
            sig_in = sig;
        }
        if (valid && q == 0) {
            const double2 b = a.cb[(int64_t)lo * a.ns + s];
            const double2 t = a.tot[s];
            const double dre = b.x + re - t.x, dim = b.y + im - t.y;
            const double mag = exp(dre) * (double)it.coef;
            a.contrib[(int64_t)it.slot * a.ns + s] = make_double2(mag * cos(dim), mag * sin(dim));
        }
    }
}

}  // namespace rnnwf
