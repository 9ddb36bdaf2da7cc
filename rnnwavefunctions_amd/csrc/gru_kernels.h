// gru_kernels.h - the two recurrent kernels of the positive-RNN (pRNN) path.
//
//   prnn_base_kernel : one pass over all N sites for every chain (16 chains per wave): ancestral
//                      sampling (1DTFIM/RNNwavefunction.py:35-74) or teacher-forced evaluation
//                      (:76-118); optionally checkpoints the hidden state after every site and the
//                      "flip base" log-probabilities that let the flip pass skip the shared prefix.
//   prnn_flip_kernel : for every (sample s, flipped site i) re-evaluates sites i+1..N-1 only, starting
//                      from the checkpoint of site i - the off-diagonal term of
//                      Ising_local_energies (1DTFIM/TrainingRNN_1DTFIM.py:42-48,56-65) without ever
//                      materialising queue_samples.
//
// Device layouts (all coalesced per wave):
//   bits [W = ceil(N/32)][ns] u32   spin n of chain s = bit (n & 31) of bits[n >> 5][s]
//   hck  [N-1][nsb][KT][64]   T     hidden state after site n, in B-fragment order (lane-linear)
//   lpq  [N+1][ns]            f64   row 0: log P(s); row k+1: log P(s with site k flipped)
#pragma once
#include "gru_core.h"

namespace rnnwf {

struct PrnnArgs {
    const void* wimg;            // packed weight image (GruLayout)
    int32_t N;                   // sites
    int64_t ns;                  // chains in this launch
    int64_t nsb;                 // ceil(ns / 16)
    uint32_t* bits;              // in (teacher) / out (sampling)
    void* hck;                   // nullptr: no checkpoints
    double* lpq;                 // nullptr: no flip base
    double* out_lp;              // [ns] log P of the chain (may be nullptr)
    const int32_t* row_of_pos;   // lpq row (1-based site) of chain position n; nullptr: n + 1
    uint64_t seed, step;
    int64_t sample_offset;
    int32_t sampling;            // 1: draw spins, 0: read them from bits
    int64_t ntiles;              // flip pass: (N-1) * nsb
    int32_t ablate;              // diagnostics only (RNNWF_ABLATE): 1 skip MFMAs, 2 skip gate arithmetic, 4 skip head
};


template <typename T, int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) prnn_base_kernel(PrnnArgs a) {
    using C = GruCore<T, NFULL, 1>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, a.wimg);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    for (int64_t sb = gw; sb < a.nsb; sb += nw) {
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        T h[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) h[kt] = T(0);
        int sig_in = -1;
        uint32_t word = 0;
        double cum = 0.0;
        for (int n = 0; n < N; ++n) {
            if (!a.sampling && (n & 31) == 0) word = a.bits[(int64_t)(n >> 5) * a.ns + sc];
            C::step(lds, sig_in, h, lane);
            T z[1];
            C::head(lds, h, lane, z);
            T lp0, lp1;
            log_softmax2(z[0], lp0, lp1);
            int sig;
            if (a.sampling) {
                // tf.multinomial(log p): class 0 iff u * total < p0
                const float u = philox_uniform(a.seed, a.step, (uint64_t)(a.sample_offset + sc), n);
                sig = ((T)u < prob0(z[0])) ? 0 : 1;
                word |= (uint32_t)sig << (n & 31);
                if (((n & 31) == 31 || n == N - 1) && valid && q == 0) a.bits[(int64_t)(n >> 5) * a.ns + s] = word;
                if ((n & 31) == 31) word = 0;
            } else {
                sig = (word >> (n & 31)) & 1;
            }
            const double lsel = (double)(sig ? lp1 : lp0);
            if (a.lpq) {
                const double loth = (double)(sig ? lp0 : lp1);
                const int64_t row = a.row_of_pos ? a.row_of_pos[n] : n + 1;
                if (valid && q == 0) a.lpq[row * a.ns + s] = cum + loth;
            }
            cum += lsel;
            if (a.hck && n < N - 1) {
                T* dst = reinterpret_cast<T*>(a.hck) + (((int64_t)n * a.nsb + sb) * KT) * 64 + lane;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) dst[kt * 64] = h[kt];
            }
            sig_in = sig;
        }
        if (valid && q == 0) {
            if (a.lpq) a.lpq[s] = cum;
            if (a.out_lp) a.out_lp[s] = cum;
        }
    }
}

template <typename T, int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) prnn_flip_kernel(PrnnArgs a) {
    using C = GruCore<T, NFULL, 1>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, a.wimg);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    // tiles are ordered longest chain first (i ascending); every wave strides through them, so each
    // wave receives the same mix of lengths
    for (int64_t tile = gw; tile < a.ntiles; tile += nw) {
        const int i = (int)(tile / a.nsb);
        const int64_t sb = tile - (int64_t)i * a.nsb;
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        T h[KT];
        {
            const T* src = reinterpret_cast<const T*>(a.hck) + (((int64_t)i * a.nsb + sb) * KT) * 64 + lane;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) h[kt] = src[kt * 64];
        }
        // Site n consumes spin n-1 and its head needs spin n: one coalesced 4-byte load per site, fetched a site
        // ahead (branch-free loop body, so the scheduler can interleave MFMA and VALU work across the whole step).
        auto spin = [&](int n) { return (int)((a.bits[(int64_t)(n >> 5) * a.ns + sc] >> (n & 31)) & 1); };
        int sig_in = 1 - spin(i);                  // the flipped spin feeds site i+1
        double lp = 0.0;
        for (int n = i + 1; n < N; ++n) {
            const int sig = spin(n);
            C::step(lds, sig_in, h, lane, a.ablate);
            if (!(a.ablate & 4)) {
                T z[1];
                C::head(lds, h, lane, z);
                T lp0, lp1;
                log_softmax2(z[0], lp0, lp1);
                lp += (double)(sig ? lp1 : lp0);
            }
            sig_in = sig;
        }
        if (valid && q == 0) {
            const int64_t row = a.row_of_pos ? a.row_of_pos[i] : i + 1;
            a.lpq[row * a.ns + s] += lp;
        }
    }
}

}  // namespace rnnwf
