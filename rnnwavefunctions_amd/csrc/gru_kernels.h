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
#include "ml_coop.h"
#include "split_core.h"

namespace rnnwf {

struct PrnnArgs {
    const void* wimg;            // packed weight image (GruLayout)
    const void* wbf;             // bf16x3 A fragments of the cooperative base pass (BaseBfLayout), nullptr: none
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
    int32_t ablate;              // diagnostics only (RNNWF_ABLATE): 1 skip MFMAs, 2 skip gate arithmetic, 4 skip head;
                                 // base pass: 8 no checkpoint stores, 16 no flip-base stores, 32 constant uniform
    unsigned long long* stamps;  // diagnostics builds only (RNNWF_STAMPS): per wave 8 counters, see prnn_flip_pp_kernel
};


template <typename T, int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) prnn_base_kernel(PrnnArgs a) {
    using C = GruCore<T, NFULL, 1>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const char* img = C::stage(lds, a.wimg);       // LDS, or the global image where it exceeds LDS (GruLayout::SPILL)
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
            C::template step<true>(img, sig_in, h, lane);
            T z[1];
            C::head(img, h, lane, z);
            T lp0, lp1;
            log_softmax2(z[0], lp0, lp1);
            int sig;
            if (a.sampling) {
                // tf.multinomial(log p): class 0 iff u * total < p0
                const float u = RNNWF_ABLATED(a.ablate, 32) ? 0.5f : philox_uniform(a.seed, a.step, (uint64_t)(a.sample_offset + sc), n);
                sig = ((T)u < prob0(z[0])) ? 0 : 1;
                word |= (uint32_t)sig << (n & 31);
                if (((n & 31) == 31 || n == N - 1) && valid && q == 0) a.bits[(int64_t)(n >> 5) * a.ns + s] = word;
                if ((n & 31) == 31) word = 0;
            } else {
                sig = (word >> (n & 31)) & 1;
            }
            const double lsel = (double)(sig ? lp1 : lp0);
            if (a.lpq && !RNNWF_ABLATED(a.ablate, 16)) {
                const double loth = (double)(sig ? lp0 : lp1);
                const int64_t row = a.row_of_pos ? a.row_of_pos[n] : n + 1;
                if (valid && q == 0) a.lpq[row * a.ns + s] = cum + loth;
            }
            cum += lsel;
            if (a.hck && n < N - 1 && !RNNWF_ABLATED(a.ablate, 8)) {
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
    const char* img = C::stage(lds, a.wimg);       // LDS, or the global image where it exceeds LDS (GruLayout::SPILL)
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
            C::step(img, sig_in, h, lane, a.ablate);
            if (!RNNWF_ABLATED(a.ablate, 4)) {
                T z[1];
                C::head(img, h, lane, z);
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

// prnn_base_coop_kernel : the base pass when there are fewer 16-chain blocks than SIMDs (config 2: 625 blocks, 1024
// SIMDs).  There the one-wave-per-block kernel is pure latency: N sequential steps of ~130 dependent-issue MFMAs on
// 60 % of the SIMDs.  Here NFULL + 1 waves share one block of 16 chains: wave m < NFULL computes the three gate tiles
// of units 16m..16m+15 (39 MFMAs at 50 units), the last wave the mixed tile of the remainder units (13 MFMAs) AND the
// head, the sampling decision and the log-probability bookkeeping.  Per site:
//     all waves  : products of their tiles with the state (no bias yet: the input spin is still being drawn)
//     barrier B  : the input spin of this site is published              (skipped at site 0)
//     all waves  : + bias / one-hot rows, gates, publish the new values of their units
//     barrier A  : the new state is complete; every wave reads it (it is everybody's next B operand)
//     last wave  : head, draw / read the spin, publish it, log-probabilities
// The arithmetic per unit is the instruction sequence of prnn_base_kernel (GruCore::step<BIAS_LAST>), so both kernels
// agree bit for bit, draws included - a batch and its shards may take different kernels.
// The cooperative site loop shared by the positive and the complex RNN: `begin(sb)` on every wave at the start of a
// block of 16 chains, `site(n, h) -> spin` on the remainder wave once the state after site n is complete (head, draw,
// bookkeeping), `end(sb)` on the remainder wave after the last site.
template <int NFULL, int NOUT, typename Begin, typename Site, typename End>
__device__ __forceinline__ void coop_base_pass(char* lds, const void* wimg, int N, int64_t nsb, void* hck, int ablate,
                                               Begin begin, Site site, End end) {
    using C = GruCore<float, NFULL, NOUT>;
    using L = typename C::L;
    using V4 = typename C::V4;
    constexpr int KT = C::KT, NG = C::NG;
    C::stage(lds, wimg);
    float* xbuf = reinterpret_cast<float*>(lds + L::BYTES);            // [2][KT][64] new state, then [2][64] int spins
    int* sbuf = reinterpret_cast<int*>(xbuf + 2 * KT * 64);
    const int lane = threadIdx.x & 63, q = lane >> 4;
    const int m = threadIdx.x >> 6;                                     // this wave's unit block (NFULL: remainder units)
    const bool full = m < NFULL;
    const V4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int64_t sb = blockIdx.x; sb < nsb; sb += gridDim.x) {
        begin(sb);
        float h[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) h[kt] = 0.0f;
        float own[4] = {0.0f, 0.0f, 0.0f, 0.0f};                        // this wave's units of the current state
        for (int n = 0; n < N; ++n) {
            asm volatile("" ::: "memory");
            float* xb = xbuf + (size_t)(n & 1) * KT * 64 + lane;
            const V4* av = reinterpret_cast<const V4*>(lds + L::OFF_AVEC) + lane;
            const float* ar = reinterpret_cast<const float*>(lds + L::OFF_AREM) + lane;
            V4 accr = zero4, accu = zero4, accq = zero4;                // remainder wave: accr only
            const int tr = full ? m : 3 * NFULL, tu = NFULL + m, tq = 2 * NFULL + m;
            if (full) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const V4 fr = av[(tr * NG + g) * 64], fu = av[(tu * NG + g) * 64], fq = av[(tq * NG + g) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        accr = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[j], h[g * 4 + j], accr, 0, 0, 0);
                        accu = __builtin_amdgcn_mfma_f32_16x16x4f32(fu[j], h[g * 4 + j], accu, 0, 0, 0);
                        accq = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[j], h[g * 4 + j], accq, 0, 0, 0);
                    }
                }
                accr = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tr * 64], h[KT - 1], accr, 0, 0, 0);
                accu = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tu * 64], h[KT - 1], accu, 0, 0, 0);
                accq = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tq * 64], h[KT - 1], accq, 0, 0, 0);
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const V4 f = av[(tr * NG + g) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) accr = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j], h[g * 4 + j], accr, 0, 0, 0);
                }
                accr = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tr * 64], h[KT - 1], accr, 0, 0, 0);
            }
            int sig_in = -1;
            if (n > 0) {
                __syncthreads();                                        // barrier B: spin of site n-1 is published
                sig_in = sbuf[((n - 1) & 1) * 64 + lane];
            }
            const char* binit = lds + L::OFF_BINIT + (size_t)(sig_in + 1) * L::SZ_BINIT_VARIANT + (size_t)q * 16;
            const char* xcp = lds + L::OFF_XC + (size_t)(sig_in + 1) * L::SZ_XC_VARIANT + (size_t)q * 16;
            if (full) {
                accr += *reinterpret_cast<const V4*>(binit + (size_t)tr * 64);
                accu += *reinterpret_cast<const V4*>(binit + (size_t)tu * 64);
                accq += *reinterpret_cast<const V4*>(binit + (size_t)tq * 64);
                const V4 xc = *reinterpret_cast<const V4*>(xcp + (size_t)m * 64);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    own[r] = gru_gate<float>(accr[r], accu[r], accq[r], xc[r], own[r]);
                    xb[(4 * m + r) * 64] = own[r];
                }
            } else {
                accr += *reinterpret_cast<const V4*>(binit + (size_t)tr * 64);
                const float xc = *reinterpret_cast<const float*>(xcp + (size_t)NFULL * 64);
                own[0] = gru_gate<float>(accr[0], accr[1], accr[2], xc, own[0]);
                xb[(KT - 1) * 64] = own[0];
            }
            if (hck && n < N - 1 && !RNNWF_ABLATED(ablate, 8)) {
                float* dst = reinterpret_cast<float*>(hck) + (((int64_t)n * nsb + sb) * KT) * 64 + lane;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(4 * m + r) * 64] = own[r];
                } else {
                    dst[(KT - 1) * 64] = own[0];
                }
            }
            __syncthreads();                                            // barrier A: the new state is complete
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) h[kt] = xb[kt * 64];
            if (!full) site(n, h, [&](int sg) { sbuf[(n & 1) * 64 + lane] = sg; });   // head + spin + bookkeeping: remainder wave only
        }
        if (!full) end(sb);
        __syncthreads();          // the next block of chains starts writing the buffers again
    }
}

// The cooperative pass with its matrix products on the bf16 matrix core (v_mfma_f32_16x16x32_bf16, bf16x3 operands: f32-accurate,
// split_core.h).  Why: on the CUs that hold three blocks of 16 chains (config 2: 625 blocks on 256 CUs) the pass above is bound by
// 3 x (39 f32-input MFMAs x 32 cycles + VALU) per SIMD and step - the f32-input MFMA runs at the vector rate and overlaps nothing.
// Here wave m's three gate tiles cost 3 tiles x 6 products x NKS k-steps = 36 MFMAs of 16 cycles, and a bf16 MFMA of one wave runs
// beside the VALU work of another.  Same tiles, rows, unit assignment, tables, gate arithmetic, head and `site` as above; what
// changes is where the B operand comes from: every wave splits the (up to four) units it has just produced into three bf16 parts
// and stores them - one 8-byte store per part - into the block's operand buffer PB [part][octet][chain] (layout.h: BaseBfLayout),
// from which every wave of the block reads its B quads after barrier A.  NB blocks of 16 chains share one workgroup (one image
// in LDS): slot b of workgroup k takes block  round x NB x grid + b x grid + k.  All waves of a workgroup run the same number of
// barriers; slots without a block idle through them.
template <int NFULL, int NOUT, typename Begin, typename Site, typename End>
__device__ __forceinline__ void coop_base_pass_bf(char* lds, const void* wimg, const void* wbf, int N, int64_t nsb, void* hck,
                                                  Begin begin, Site site, End end, unsigned long long* stamps = nullptr) {
    using C = GruCore<float, NFULL, NOUT>;
    using L = typename C::L;
    using B = BaseBfLayout<NFULL>;
    using V4 = typename C::V4;
    constexpr int KT = C::KT, NW = B::NW, NKS = B::NKS, NO = B::NO, NB = B::NB;
    constexpr int NWB = NW + 1;                                        // waves per block: NW product / gate waves + the sampler
    constexpr size_t SLOT = B::PB_BYTES + (size_t)2 * KT * 64 * 4 + 2 * 64 * 4;
    {   // stage: the tables of the f32 image (its A fragments are not used here) and the bf16 A fragments behind it
        const uint4* s0 = reinterpret_cast<const uint4*>(wimg);
        uint4* d0 = reinterpret_cast<uint4*>(lds);
        for (int i = (int)(L::OFF_BINIT / 16) + threadIdx.x; i < (int)(L::BYTES / 16); i += blockDim.x) d0[i] = s0[i];
        const uint4* s1 = reinterpret_cast<const uint4*>(wbf);
        uint4* d1 = reinterpret_cast<uint4*>(lds + L::BYTES);
        for (int i = threadIdx.x; i < (int)(B::BYTES / 16); i += blockDim.x) d1[i] = s1[i];
        uint4* z = reinterpret_cast<uint4*>(lds + L::BYTES + B::BYTES);        // operand buffers start as zeros (padding entries stay zero)
        for (int i = threadIdx.x; i < (int)(NB * SLOT / 16); i += blockDim.x) z[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = wave / NWB, m = wave - b * NWB;                      // block slot; unit group (NFULL: remainder units), NW: the sampler
    const bool full = m < NFULL, sampler = m == NW;
    char* slot = lds + L::BYTES + B::BYTES + (size_t)b * SLOT;
    char* pb = slot;
    float* xbuf = reinterpret_cast<float*>(slot + B::PB_BYTES);        // [2][KT][64] new state (f32), then [2][64] int spins
    int* sbuf = reinterpret_cast<int*>(xbuf + 2 * KT * 64);
    const u32x4* abf = reinterpret_cast<const u32x4*>(lds + L::BYTES) + lane;
    const u32x4* bq = reinterpret_cast<const u32x4*>(pb) + (q * 16 + c);      // B quad (part p, k-step t) at [(p NO + 4 t) * 16]
    const V4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int ORD[6][2] = {{2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};      // (weight part, state part), smallest products first
#ifdef RNNWF_DIAGNOSTICS      // in-kernel cycle stamps (tools/stamps_base.py): where a site's cycles go, per wave role
    unsigned long long tseg[5] = {0, 0, 0, 0, 0}, ts = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = ts, r_begin = __builtin_amdgcn_s_memrealtime();
#define RNNWF_BSTAMP(k_) do { if (stamps) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tseg[k_] += t_ - ts; ts = t_; } } while (0)
#else
#define RNNWF_BSTAMP(k_) do { } while (0)
#endif
    for (int64_t base = 0; base < nsb; base += (int64_t)gridDim.x * NB) {
        const int64_t sb = base + (int64_t)b * gridDim.x + blockIdx.x;
        const bool active = sb < nsb;
        if (active) begin(sb);
        float own[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int n = 0; n < N; ++n) {
            asm volatile("" ::: "memory");
            float* xb = xbuf + (size_t)(n & 1) * KT * 64 + lane;
            V4 accr = zero4, accu = zero4, accq = zero4;
            const int tr = full ? m : 3 * NFULL, tu = NFULL + m, tq = 2 * NFULL + m;
            if (active && !sampler && n > 0) {                          // (site 0: the state is zero)
                u32x4 hb[3][NKS];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int t = 0; t < NKS; ++t) hb[p][t] = bq[(p * NO + 4 * t) * 16];
                if (full) {
#pragma unroll
                    for (int o = 0; o < 6; ++o)
#pragma unroll
                        for (int t = 0; t < NKS; ++t) {
                            const bf16x8 bb = __builtin_bit_cast(bf16x8, hb[ORD[o][1]][t]);
                            const int w = ORD[o][0];
                            accr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, abf[((tr * 3 + w) * NKS + t) * 64]), bb, accr, 0, 0, 0);
                            accu = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, abf[((tu * 3 + w) * NKS + t) * 64]), bb, accu, 0, 0, 0);
                            accq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, abf[((tq * 3 + w) * NKS + t) * 64]), bb, accq, 0, 0, 0);
                        }
                } else {
#pragma unroll
                    for (int o = 0; o < 6; ++o)
#pragma unroll
                        for (int t = 0; t < NKS; ++t)
                            accr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, abf[((tr * 3 + ORD[o][0]) * NKS + t) * 64]),
                                                                           __builtin_bit_cast(bf16x8, hb[ORD[o][1]][t]), accr, 0, 0, 0);
                }
            }
            RNNWF_BSTAMP(0);
            __syncthreads();                                            // barrier B: spin of site n-1 is published, PB has been read
            RNNWF_BSTAMP(1);
            const int sig_in = n > 0 ? sbuf[((n - 1) & 1) * 64 + lane] : -1;
            if (active && !sampler) {
                const char* binit = lds + L::OFF_BINIT + (size_t)(sig_in + 1) * L::SZ_BINIT_VARIANT + (size_t)q * 16;
                const char* xcp = lds + L::OFF_XC + (size_t)(sig_in + 1) * L::SZ_XC_VARIANT + (size_t)q * 16;
                if (full) {
                    accr += *reinterpret_cast<const V4*>(binit + (size_t)tr * 64);
                    accu += *reinterpret_cast<const V4*>(binit + (size_t)tu * 64);
                    accq += *reinterpret_cast<const V4*>(binit + (size_t)tq * 64);
                    const V4 xc = *reinterpret_cast<const V4*>(xcp + (size_t)m * 64);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        own[r] = gru_gate<float>(accr[r], accu[r], accq[r], xc[r], own[r]);
                        xb[(4 * m + r) * 64] = own[r];
                    }
                } else {
                    accr += *reinterpret_cast<const V4*>(binit + (size_t)tr * 64);
                    const float xc = *reinterpret_cast<const float*>(xcp + (size_t)NFULL * 64);
                    own[0] = gru_gate<float>(accr[0], accr[1], accr[2], xc, own[0]);
                    xb[(KT - 1) * 64] = own[0];
                }
                if (hck && n < N - 1) {
                    float* dst = reinterpret_cast<float*>(hck) + (((int64_t)n * nsb + sb) * KT) * 64 + lane;
                    if (full) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dst[(4 * m + r) * 64] = own[r];
                    } else {
                        dst[(KT - 1) * 64] = own[0];
                    }
                }
                // the new values as three exact bf16 parts each, into this lane's entries 4 q .. 4 q + 3 of group m
                {
                    const unsigned p1a = cvt_pk_bf16(own[0], own[1]), p1b = cvt_pk_bf16(own[2], own[3]);
                    const float r0 = own[0] - __uint_as_float(p1a << 16), r1 = own[1] - __uint_as_float(p1a & 0xffff0000u);
                    const float r2 = own[2] - __uint_as_float(p1b << 16), r3 = own[3] - __uint_as_float(p1b & 0xffff0000u);
                    const unsigned p2a = cvt_pk_bf16(r0, r1), p2b = cvt_pk_bf16(r2, r3);
                    const float s0 = r0 - __uint_as_float(p2a << 16), s1 = r1 - __uint_as_float(p2a & 0xffff0000u);
                    const float s2 = r2 - __uint_as_float(p2b << 16), s3 = r3 - __uint_as_float(p2b & 0xffff0000u);
                    const unsigned p3a = cvt_pk_bf16(s0, s1), p3b = cvt_pk_bf16(s2, s3);
                    char* dst = pb + ((size_t)(2 * m + (q >> 1)) * 16 + c) * 16 + 8 * (q & 1);
                    if (full) {
                        *reinterpret_cast<uint2*>(dst) = make_uint2(p1a, p1b);
                        *reinterpret_cast<uint2*>(dst + (size_t)NO * 256) = make_uint2(p2a, p2b);
                        *reinterpret_cast<uint2*>(dst + (size_t)2 * NO * 256) = make_uint2(p3a, p3b);
                    } else {                                            // one unit per lane: entry 4 q of the remainder group
                        *reinterpret_cast<unsigned short*>(dst) = (unsigned short)(p1a & 0xffffu);
                        *reinterpret_cast<unsigned short*>(dst + (size_t)NO * 256) = (unsigned short)(p2a & 0xffffu);
                        *reinterpret_cast<unsigned short*>(dst + (size_t)2 * NO * 256) = (unsigned short)(p3a & 0xffffu);
                    }
                }
            }
            RNNWF_BSTAMP(2);
            __syncthreads();                                            // barrier A: the new state (f32 and parts) is complete
            RNNWF_BSTAMP(3);
            if (sampler && active) {                                    // head + spin (published at once) + bookkeeping
                float h[KT];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) h[kt] = xb[kt * 64];
                site(n, h, [&](int sg) { sbuf[(n & 1) * 64 + lane] = sg; });
            }
            RNNWF_BSTAMP(4);
        }
        if (sampler && active) end(sb);
        __syncthreads();          // the next round starts writing the buffers again
    }
#ifdef RNNWF_DIAGNOSTICS
    if (stamps && lane == 0) {
        unsigned long long* o = stamps + ((int64_t)blockIdx.x * (NB * NWB) + wave) * 8;
        for (int k = 0; k < 5; ++k) o[k] = tseg[k];
        o[5] = __builtin_amdgcn_s_memtime() - t_begin;
        o[6] = __builtin_amdgcn_s_memrealtime() - r_begin;
        o[7] = (unsigned long long)m;
    }
#endif
}

// NL > 1: stacked layers (ml_coop.h), same hooks
template <int NFULL, bool BF = false, int NL = 1>
__global__ void __launch_bounds__((NL > 1 ? MlCoopLayout<NFULL, NL, 1>::THREADS : BF ? (NFULL + 2) * 64 * BaseBfLayout<NFULL>::NB : (NFULL + 1) * 64))
prnn_base_coop_kernel(PrnnArgs a) {
    using C = GruCore<float, NFULL, 1>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int N = a.N;
    int64_t s = 0, sc = 0;
    bool valid = false;
    uint32_t word = 0, word_in = 0;
    double cum = 0.0;
    float u_next = 0.0f;
    // what site n + 1 needs that does not depend on the state - its uniform (sampling) or its word of given spins (teacher
    // forcing) - is fetched at the END of site n, behind the publication of site n's spin: off the per-site critical path
    auto prefetch = [&](int n) {
        if (a.sampling) u_next = RNNWF_ABLATED(a.ablate, 32) ? 0.5f : philox_uniform(a.seed, a.step, (uint64_t)(a.sample_offset + sc), n);
        else if ((n & 31) == 0) word_in = a.bits[(int64_t)(n >> 5) * a.ns + sc];
    };
    auto begin = [&](int64_t sb) {
            s = sb * kChains + c;
            valid = s < a.ns;
            sc = valid ? s : a.ns - 1;
            word = 0;
            cum = 0.0;
            prefetch(0);
        };
    auto site =
        [&](int n, const float (&h)[KT], auto&& publish) {
            float zz[1];
            C::head(lds, h, lane, zz);
            const float z = zz[0];
            int sig;
            if (a.sampling) sig = (u_next < prob0(z)) ? 0 : 1;
            else sig = (word_in >> (n & 31)) & 1;
            publish(sig);                                               // the other waves wait for this
            float lp0, lp1;
            log_softmax2(z, lp0, lp1);
            if (a.sampling) {
                word |= (uint32_t)sig << (n & 31);
                if (((n & 31) == 31 || n == N - 1) && valid && q == 0) a.bits[(int64_t)(n >> 5) * a.ns + s] = word;
                if ((n & 31) == 31) word = 0;
            }
            if (a.lpq && !RNNWF_ABLATED(a.ablate, 16)) {
                const int64_t row = a.row_of_pos ? a.row_of_pos[n] : n + 1;
                if (valid && q == 0) a.lpq[row * a.ns + s] = cum + (double)(sig ? lp0 : lp1);
            }
            cum += (double)(sig ? lp1 : lp0);
            if (n + 1 < N) prefetch(n + 1);
        };
    auto end = [&](int64_t) {
            if (valid && q == 0) {
                if (a.lpq) a.lpq[s] = cum;
                if (a.out_lp) a.out_lp[s] = cum;
            }
        };
    if constexpr (NL > 1) coop_ml_base_pass<NFULL, NL, 1>(lds, a.wimg, N, a.nsb, a.hck, begin, site, end);
    else if constexpr (BF) coop_base_pass_bf<NFULL, 1>(lds, a.wimg, a.wbf, N, a.nsb, a.hck, begin, site, end, a.stamps);
    else coop_base_pass<NFULL, 1>(lds, a.wimg, N, a.nsb, a.hck, a.ablate, begin, site, end);
}

}  // namespace rnnwf
