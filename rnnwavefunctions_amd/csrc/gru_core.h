// gru_core.h - one recurrent step of the cuDNN-compatible GRU for 16 chains per wave, on MFMA.
//
// Reference arithmetic: tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell.call as invoked at
// 1DTFIM/RNNwavefunction.py:66,108 and J1J2/ComplexRNNwavefunction.py:80,140:
//     g = sigmoid([x,h] Wg + bg);  r,u = split(g)
//     c = tanh(x Wci + bci + r * (h Wch + bch));   h' = (1-u) c + u h
// x is a one-hot (or the zero vector at the first site), so its matmuls are row selections that are
// folded into the accumulator initialisation (tables BINIT / XC of layout.h).
#pragma once
#include "device.h"

namespace rnnwf {

template <typename T, int NFULL, int NOUT>
struct GruCore {
    using L = GruLayout<T, NFULL, NOUT>;
    using F = Frag<T>;
    using V4 = typename F::V4;
    using VA = typename F::VA;
    static constexpr int KT = L::KT, NT = L::NT, NG = L::NG, VW = L::VW;
    static constexpr int TC = 5;  // tiles per MFMA issue group (independent accumulators back to back)

    // Stage the packed weight image into LDS (all threads of the workgroup).
    static __device__ __forceinline__ void stage(char* lds, const void* wimg) {
        const uint4* src = reinterpret_cast<const uint4*>(wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < (int)(L::BYTES / 16); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }

    // h[kt] of lane (c, q) holds unit 4 kt + q of chain c.  sig: input spin of this step (-1: zero vector).
    static __device__ __forceinline__ void step(const char* lds, int sig, T (&h)[KT], int lane) {
        const int q = lane >> 4;
        // The weight image never changes, so the compiler would hoist all ~NT*KT fragment loads out of
        // the site loop and pin them in registers (1 wave/SIMD).  Re-read them from LDS every step.
        asm volatile("" ::: "memory");
        V4 acc[NT];
        {
            const char* b = lds + L::OFF_BINIT + (size_t)(sig + 1) * L::SZ_BINIT_VARIANT + (size_t)q * 4 * sizeof(T);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = *reinterpret_cast<const V4*>(b + (size_t)t * 16 * sizeof(T));
        }
        const VA* av = reinterpret_cast<const VA*>(lds + L::OFF_AVEC) + lane;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int t0 = 0; t0 < NT; t0 += TC) {
                VA a[TC];
#pragma unroll
                for (int t = 0; t < TC; ++t)
                    if (t0 + t < NT) a[t] = av[((t0 + t) * NG + g) * 64];
#pragma unroll
                for (int j = 0; j < VW; ++j)
#pragma unroll
                    for (int t = 0; t < TC; ++t)
                        if (t0 + t < NT) acc[t0 + t] = F::mfma(a[t][j], h[g * VW + j], acc[t0 + t]);
            }
        }
        {
            const T* ar = reinterpret_cast<const T*>(lds + L::OFF_AREM) + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = F::mfma(ar[t * 64], h[KT - 1], acc[t]);
        }
        const char* x = lds + L::OFF_XC + (size_t)(sig + 1) * L::SZ_XC_VARIANT + (size_t)q * 4 * sizeof(T);
#pragma unroll
        for (int m = 0; m < NFULL; ++m) {
            const V4 xc = *reinterpret_cast<const V4*>(x + (size_t)m * 16 * sizeof(T));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T rg = sigmoid_(acc[m][r]);
                const T ug = sigmoid_(acc[NFULL + m][r]);
                const T cc = tanh_(xc[r] + rg * acc[2 * NFULL + m][r]);
                h[4 * m + r] = (T(1) - ug) * cc + ug * h[4 * m + r];
            }
        }
        {
            const V4 xc = *reinterpret_cast<const V4*>(x + (size_t)NFULL * 16 * sizeof(T));
            const V4 a = acc[NT - 1];
            const T rg = sigmoid_(a[0]);
            const T ug = sigmoid_(a[1]);
            const T cc = tanh_(xc[0] + rg * a[2]);
            h[KT - 1] = (T(1) - ug) * cc + ug * h[KT - 1];
        }
    }

    // Dense(NOUT) on the new hidden state: z = h' Wd + bd, reduced over the four lane quarters.
    // (tf.layers.Dense at 1DTFIM/RNNwavefunction.py:33,67,109; two heads for the cRNN, :42-43.)
    static __device__ __forceinline__ void head(const char* lds, const T (&h)[KT], int lane, T (&z)[NOUT]) {
        const int q = lane >> 4;
        asm volatile("" ::: "memory");
        const T* wd = reinterpret_cast<const T*>(lds + L::OFF_WD) + q * NOUT;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) z[o] = T(0);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) z[o] += h[kt] * wd[kt * 4 * NOUT + o];
        const T* bd = reinterpret_cast<const T*>(lds + L::OFF_BD);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            z[o] += __shfl_xor(z[o], 16);
            z[o] += __shfl_xor(z[o], 32);
            z[o] += bd[o];
        }
    }

    // softmax over two logits as tf.nn.softmax computes it: exp(z - max) / sum.
    static __device__ __forceinline__ void softmax2(T z0, T z1, T& p0, T& p1) {
        const T m = z0 > z1 ? z0 : z1;
        const T e0 = exp_(z0 - m), e1 = exp_(z1 - m);
        const T s = e0 + e1;
        p0 = e0 / s;
        p1 = e1 / s;
    }
};

}  // namespace rnnwf
