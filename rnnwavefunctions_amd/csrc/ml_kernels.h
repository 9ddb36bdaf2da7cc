// ml_kernels.h - base and flip pass of the positive GRU RNN with NL > 1 stacked layers
// (units = [h] * num_layers, 1DTFIM/TrainingRNN_1DTFIM.py:98; MultiRNNCell at 1DTFIM/RNNwavefunction.py:32).
//
// Same decomposition as gru_kernels.h.  Every layer's state stays in registers in B-fragment order; the new
// state of layer l is directly the X operand of layer l+1 (gru_core.h, UpperCore), the head reads the top layer.
// T = float (1D positive RNN) or double (2DTFIM_1DRNN, units=[num_units]*num_layers at Training1DRNN_2DTFIM.py:94).
// LDS image: [GruLayout<T, NFULL, 1> | UpperLayout<NFULL, T> x (NL-1)].
//   hck [N][nsb][NL][KT][64] T     states of all layers after site n (all N sites: the layer-wise gradient of
//                                  ml_grad_kernels.h needs the last site's lower-layer states too)
#pragma once
#include "gru_kernels.h"

namespace rnnwf {

template <int NFULL, int NL, typename T = float>
struct MlCore {
    using C0 = GruCore<T, NFULL, 1>;
    using CU = UpperCore<NFULL, T>;
    static constexpr int KT = C0::KT;
    static constexpr int SPILL = MlSpill<NFULL, NL, T>::value;
    static constexpr size_t BYTES = C0::L::BYTES + (size_t)(NL - 1 - SPILL) * CU::U::BYTES;     // LDS-resident part

    static __device__ __forceinline__ void stage(char* lds, const void* wimg) {
        const uint4* src = reinterpret_cast<const uint4*>(wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < (int)(BYTES / 16); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    // all layers for one site; returns the head logit difference of the top layer
    static __device__ __forceinline__ T step(const char* lds, const void* wimg, int sig_in, T (&h)[NL][KT], int lane) {
        C0::step_plain(lds, sig_in, h[0], lane);
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            const size_t off = C0::L::BYTES + (size_t)(l - 1) * CU::U::BYTES;
            if (l < NL - SPILL) CU::step(lds + off, h[l - 1], h[l], lane);
            else CU::step(reinterpret_cast<const char*>(wimg) + off, h[l - 1], h[l], lane, NFULL * (int)sizeof(T) >= 16);      // wide layers (f32 >= 53, f64 >= 37 units): bounded look-ahead (UpperCore::block)
        }
        T z[1];
        C0::head(lds, h[NL - 1], lane, z);
        return z[0];
    }
};

template <typename T, int NFULL, int NL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) prnn_ml_base_kernel(PrnnArgs a) {
    using M = MlCore<NFULL, NL, T>;
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
        T h[NL][KT];
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) h[l][kt] = T(0);
        int sig_in = -1;
        uint32_t word = 0;
        double cum = 0.0;
        for (int n = 0; n < N; ++n) {
            if (!a.sampling && (n & 31) == 0) word = a.bits[(int64_t)(n >> 5) * a.ns + sc];
            const T d = M::step(lds, a.wimg, sig_in, h, lane);
            T lp0, lp1;
            log_softmax2(d, lp0, lp1);
            int sig;
            if (a.sampling) {
                const float u = philox_uniform(a.seed, a.step, (uint64_t)(a.sample_offset + sc), n);
                sig = ((T)u < prob0(d)) ? 0 : 1;
                word |= (uint32_t)sig << (n & 31);
                if (((n & 31) == 31 || n == N - 1) && valid && q == 0) a.bits[(int64_t)(n >> 5) * a.ns + s] = word;
                if ((n & 31) == 31) word = 0;
            } else {
                sig = (word >> (n & 31)) & 1;
            }
            const double lsel = (double)(sig ? lp1 : lp0);
            if (a.lpq) {
                const int64_t row = a.row_of_pos ? a.row_of_pos[n] : n + 1;
                if (valid && q == 0) a.lpq[row * a.ns + s] = cum + (double)(sig ? lp0 : lp1);
            }
            cum += lsel;
            if (a.hck) {
                T* dst = reinterpret_cast<T*>(a.hck) + (((int64_t)n * a.nsb + sb) * NL * KT) * 64 + lane;
#pragma unroll
                for (int l = 0; l < NL; ++l)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) dst[(l * KT + kt) * 64] = h[l][kt];
            }
            sig_in = sig;
        }
        if (valid && q == 0) {
            if (a.lpq) a.lpq[s] = cum;
            if (a.out_lp) a.out_lp[s] = cum;
        }
    }
}

template <typename T, int NFULL, int NL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) prnn_ml_flip_kernel(PrnnArgs a) {
    using M = MlCore<NFULL, NL, T>;
    constexpr int KT = M::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    M::stage(lds, a.wimg);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    for (int64_t tile = gw; tile < a.ntiles; tile += nw) {
        const int i = (int)(tile / a.nsb);
        const int64_t sb = tile - (int64_t)i * a.nsb;
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        T h[NL][KT];
        {
            const T* src = reinterpret_cast<const T*>(a.hck) + (((int64_t)i * a.nsb + sb) * NL * KT) * 64 + lane;
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) h[l][kt] = src[(l * KT + kt) * 64];
        }
        auto spin = [&](int n) { return (int)((a.bits[(int64_t)(n >> 5) * a.ns + sc] >> (n & 31)) & 1); };
        int sig_in = 1 - spin(i);
        double lp = 0.0;
        for (int n = i + 1; n < N; ++n) {
            const int sig = spin(n);
            const T d = M::step(lds, a.wimg, sig_in, h, lane);
            T lp0, lp1;
            log_softmax2(d, lp0, lp1);
            lp += (double)(sig ? lp1 : lp0);
            sig_in = sig;
        }
        if (valid && q == 0) {
            const int64_t row = a.row_of_pos ? a.row_of_pos[i] : i + 1;
            a.lpq[row * a.ns + s] += lp;
        }
    }
}

}  // namespace rnnwf
