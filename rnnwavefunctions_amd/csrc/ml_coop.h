// ml_coop.h - the base pass of STACKED GRU layers (sampling / teacher-forced, all N sites in sequence) with the gate tiles of every
// layer split over NFULL + 1 waves per block of 16 chains - the stacked counterpart of gru_kernels.h: coop_base_pass.
//
// Why: the one-wave-per-block kernels (ml_kernels.h: prnn_ml_base_kernel, crnn_ml_kernels.h) run 130 f32-input MFMAs for the first
// layer and 260 for every layer above per site and chain block - 12 500 MFMA cycles per site with two layers of 50 units, on 625 of
// the chip's 1 024 SIMDs at config 2's size, the others idle: 0.78 ms of an 8.5 ms step (profiles/r04_z_bench_cfg2_l2.json).  Here a
// block's products are spread over four waves (= four SIMDs) and three blocks share a workgroup and its image in LDS, so every SIMD
// of the chip carries its share of the matrix work.
//
// Per site, for wave m of a block (m < NFULL: the 16 units 16 m .. 16 m + 15 of every layer; m = NFULL: the remainder units, the head
// and the draw - the Begin / Site / End hooks of coop_base_pass):
//     products of layer 0 over h_0(n-1) and the H blocks of the layers above over h_l(n-1)      (nothing of site n needed yet)
//     barrier B: the spin of site n-1 is published
//     layer 0: + bias / one-hot input rows, gates, own units -> LDS;  barrier;  every wave reads h_0(n)
//     layer l = 1 .. NL-1: X block over h_{l-1}(n), + bias, gates, own units -> LDS;  barrier;  every wave reads h_l(n)
//     remainder wave: head on h_{NL-1}(n), draw, publish the spin, bookkeeping
// LDS image [GruLayout<float, NFULL, NOUT> | UpperLayout<NFULL> x (NL - 1)] as the one-wave kernels use it (all layers resident:
// MlCoopLayout::FITS); checkpoints hck [N][nsb][NL][KT][64] as they write them.  The accumulation order differs from theirs (bias last,
// H block before X block): same values to f32 rounding, not the same bits - a handle therefore takes this pass for EVERY batch size.
#pragma once
#include "gru_core.h"

namespace rnnwf {

template <int NFULL, int NL, int NOUT>
struct MlCoopLayout {
    using L0 = GruLayout<float, NFULL, NOUT>;
    using U = UpperLayout<NFULL, float>;
    static constexpr int KT = L0::KT;
    static constexpr int NB = 3;                       // blocks of 16 chains per workgroup
    static constexpr int NWB = NFULL + 1;              // waves per block
    static constexpr int THREADS = NB * NWB * 64;
    static constexpr size_t IMG = L0::BYTES + (size_t)(NL - 1) * U::BYTES;
    static constexpr size_t SLOT = (size_t)NL * 2 * KT * 64 * 4 + 2 * 64 * 4;      // per block: [NL][2][KT][64] f32 states, [2][64] spins
    static constexpr size_t LDS = IMG + NB * SLOT;
    static constexpr bool FITS = NFULL <= 3 && LDS <= 160 * 1024;
};

template <int NFULL, int NL, int NOUT, typename Begin, typename Site, typename End>
__device__ __forceinline__ void coop_ml_base_pass(char* lds, const void* wimg, int N, int64_t nsb, void* hck, Begin begin, Site site, End end) {
    using ML = MlCoopLayout<NFULL, NL, NOUT>;
    using C0 = GruCore<float, NFULL, NOUT>;
    using L0 = typename C0::L;
    using U = typename ML::U;
    using V4 = typename C0::V4;
    constexpr int KT = ML::KT, NG = L0::NG, NB = ML::NB, NWB = ML::NWB;
    static_assert(U::NG == NG && U::KT == KT, "layers share the k-step count");
    {
        const uint4* src = reinterpret_cast<const uint4*>(wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < (int)(ML::IMG / 16); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = wave / NWB, m = wave - b * NWB;
    const bool full = m < NFULL;
    float* xbuf = reinterpret_cast<float*>(lds + ML::IMG + (size_t)b * ML::SLOT);
    int* sbuf = reinterpret_cast<int*>(xbuf + (size_t)NL * 2 * KT * 64);
    const V4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    // tiles of this wave: (r, u, third group) of its 16 units, or the mixed tile of the remainder units
    const int tr = full ? m : 3 * NFULL, tu = NFULL + m, t3 = 2 * NFULL + m;
    for (int64_t base = 0; base < nsb; base += (int64_t)gridDim.x * NB) {
        const int64_t sb = base + (int64_t)b * gridDim.x + blockIdx.x;
        const bool active = sb < nsb;
        if (active && !full) begin(sb);
        float hl[NL][KT], own[NL][4];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) hl[l][kt] = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) own[l][r] = 0.0f;
        }
        for (int n = 0; n < N; ++n) {
            asm volatile("" ::: "memory");
            V4 a0r = zero4, a0u = zero4, a0q = zero4;              // layer 0 (remainder wave: a0r = the mixed tile)
            V4 ur[NL], uu[NL], uq[NL], uy[NL];                      // layers above (index l; remainder wave: ur = the mixed tile)
#pragma unroll
            for (int l = 1; l < NL; ++l) ur[l] = uu[l] = uq[l] = uy[l] = zero4;
            if (active && n > 0) {                                  // (site 0: every state is zero)
                {
                    const V4* av = reinterpret_cast<const V4*>(lds + L0::OFF_AVEC) + lane;
                    const float* ar = reinterpret_cast<const float*>(lds + L0::OFF_AREM) + lane;
                    if (full) {
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const V4 fr = av[(tr * NG + g) * 64], fu = av[(tu * NG + g) * 64], fq = av[(t3 * NG + g) * 64];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                a0r = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[j], hl[0][g * 4 + j], a0r, 0, 0, 0);
                                a0u = __builtin_amdgcn_mfma_f32_16x16x4f32(fu[j], hl[0][g * 4 + j], a0u, 0, 0, 0);
                                a0q = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[j], hl[0][g * 4 + j], a0q, 0, 0, 0);
                            }
                        }
                        a0r = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tr * 64], hl[0][KT - 1], a0r, 0, 0, 0);
                        a0u = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tu * 64], hl[0][KT - 1], a0u, 0, 0, 0);
                        a0q = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t3 * 64], hl[0][KT - 1], a0q, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const V4 f = av[(tr * NG + g) * 64];
#pragma unroll
                            for (int j = 0; j < 4; ++j) a0r = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j], hl[0][g * 4 + j], a0r, 0, 0, 0);
                        }
                        a0r = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tr * 64], hl[0][KT - 1], a0r, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int l = 1; l < NL; ++l) {                      // H blocks: rows r, u, q over the layer's own state
                    const char* up = lds + L0::BYTES + (size_t)(l - 1) * U::BYTES;
                    const V4* av = reinterpret_cast<const V4*>(up + U::OFF_AH) + lane;
                    const float* ar = reinterpret_cast<const float*>(up + U::OFF_AHR) + lane;
                    if (full) {
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const V4 fr = av[(tr * NG + g) * 64], fu = av[(tu * NG + g) * 64], fq = av[(t3 * NG + g) * 64];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                ur[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[j], hl[l][g * 4 + j], ur[l], 0, 0, 0);
                                uu[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(fu[j], hl[l][g * 4 + j], uu[l], 0, 0, 0);
                                uq[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[j], hl[l][g * 4 + j], uq[l], 0, 0, 0);
                            }
                        }
                        ur[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tr * 64], hl[l][KT - 1], ur[l], 0, 0, 0);
                        uu[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tu * 64], hl[l][KT - 1], uu[l], 0, 0, 0);
                        uq[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t3 * 64], hl[l][KT - 1], uq[l], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const V4 f = av[(tr * NG + g) * 64];
#pragma unroll
                            for (int j = 0; j < 4; ++j) ur[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j], hl[l][g * 4 + j], ur[l], 0, 0, 0);
                        }
                        ur[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tr * 64], hl[l][KT - 1], ur[l], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                                        // barrier B: the spin of site n - 1 is published
            const int sig_in = n > 0 ? sbuf[((n - 1) & 1) * 64 + lane] : -1;
            // ---- layer 0: bias / one-hot input rows last (as coop_base_pass), gates of the wave's own units
            {
                float* xb = xbuf + (size_t)(n & 1) * KT * 64 + lane;
                if (active) {
                    const char* binit = lds + L0::OFF_BINIT + (size_t)(sig_in + 1) * L0::SZ_BINIT_VARIANT + (size_t)q * 16;
                    const char* xcp = lds + L0::OFF_XC + (size_t)(sig_in + 1) * L0::SZ_XC_VARIANT + (size_t)q * 16;
                    float* dst = hck ? reinterpret_cast<float*>(hck) + (((int64_t)n * nsb + sb) * NL * KT) * 64 + lane : nullptr;
                    if (full) {
                        a0r += *reinterpret_cast<const V4*>(binit + (size_t)tr * 64);
                        a0u += *reinterpret_cast<const V4*>(binit + (size_t)tu * 64);
                        a0q += *reinterpret_cast<const V4*>(binit + (size_t)t3 * 64);
                        const V4 xc = *reinterpret_cast<const V4*>(xcp + (size_t)m * 64);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            own[0][r] = gru_gate<float>(a0r[r], a0u[r], a0q[r], xc[r], own[0][r]);
                            xb[(4 * m + r) * 64] = own[0][r];
                            if (dst) dst[(4 * m + r) * 64] = own[0][r];
                        }
                    } else {
                        a0r += *reinterpret_cast<const V4*>(binit + (size_t)tr * 64);
                        const float xc = *reinterpret_cast<const float*>(xcp + (size_t)NFULL * 64);
                        own[0][0] = gru_gate<float>(a0r[0], a0r[1], a0r[2], xc, own[0][0]);
                        xb[(KT - 1) * 64] = own[0][0];
                        if (dst) dst[(KT - 1) * 64] = own[0][0];
                    }
                }
                __syncthreads();                                    // layer 0's new state is complete
                if (active) {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) hl[0][kt] = xb[kt * 64];
                }
            }
            // ---- layers above: X block over the new state below, bias, gates (gru_core.h: UpperCore::step)
#pragma unroll
            for (int l = 1; l < NL; ++l) {
                float* xb = xbuf + ((size_t)l * 2 + (n & 1)) * KT * 64 + lane;
                if (active) {
                    const char* up = lds + L0::BYTES + (size_t)(l - 1) * U::BYTES;
                    const V4* av = reinterpret_cast<const V4*>(up + U::OFF_AX) + lane;
                    const float* ar = reinterpret_cast<const float*>(up + U::OFF_AXR) + lane;
                    const char* bias = up + U::OFF_B + (size_t)q * 16;
                    float* dst = hck ? reinterpret_cast<float*>(hck) + ((((int64_t)n * nsb + sb) * NL + l) * KT) * 64 + lane : nullptr;
                    if (full) {
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const V4 fr = av[(tr * NG + g) * 64], fu = av[(tu * NG + g) * 64], fy = av[(t3 * NG + g) * 64];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                ur[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[j], hl[l - 1][g * 4 + j], ur[l], 0, 0, 0);
                                uu[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(fu[j], hl[l - 1][g * 4 + j], uu[l], 0, 0, 0);
                                uy[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[j], hl[l - 1][g * 4 + j], uy[l], 0, 0, 0);
                            }
                        }
                        ur[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tr * 64], hl[l - 1][KT - 1], ur[l], 0, 0, 0);
                        uu[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tu * 64], hl[l - 1][KT - 1], uu[l], 0, 0, 0);
                        uy[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t3 * 64], hl[l - 1][KT - 1], uy[l], 0, 0, 0);
                        // accumulator tiles of UpperLayout: r | u | q | y | mixed
                        ur[l] += *reinterpret_cast<const V4*>(bias + (size_t)m * 64);
                        uu[l] += *reinterpret_cast<const V4*>(bias + (size_t)(NFULL + m) * 64);
                        uq[l] += *reinterpret_cast<const V4*>(bias + (size_t)(2 * NFULL + m) * 64);
                        uy[l] += *reinterpret_cast<const V4*>(bias + (size_t)(3 * NFULL + m) * 64);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            own[l][r] = gru_gate<float>(ur[l][r], uu[l][r], uq[l][r], uy[l][r], own[l][r]);
                            xb[(4 * m + r) * 64] = own[l][r];
                            if (dst) dst[(4 * m + r) * 64] = own[l][r];
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const V4 f = av[(tr * NG + g) * 64];
#pragma unroll
                            for (int j = 0; j < 4; ++j) ur[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j], hl[l - 1][g * 4 + j], ur[l], 0, 0, 0);
                        }
                        ur[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[tr * 64], hl[l - 1][KT - 1], ur[l], 0, 0, 0);
                        ur[l] += *reinterpret_cast<const V4*>(bias + (size_t)(4 * NFULL) * 64);
                        own[l][0] = gru_gate<float>(ur[l][0], ur[l][1], ur[l][2], ur[l][3], own[l][0]);      // slots r, u, q, y
                        xb[(KT - 1) * 64] = own[l][0];
                        if (dst) dst[(KT - 1) * 64] = own[l][0];
                    }
                }
                __syncthreads();                                    // layer l's new state is complete
                if (active) {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) hl[l][kt] = xb[kt * 64];
                }
            }
            if (active && !full) site(n, hl[NL - 1], [&](int sg) { sbuf[(n & 1) * 64 + lane] = sg; });
        }
        if (active && !full) end(sb);
        __syncthreads();          // the next round starts writing the buffers again
    }
}

}  // namespace rnnwf
