// split_pp.h - the bf16x3 GRU step (split_core.h, K-packed layout MODE 2) cut into the two segments of the ping-pong
// kernels, each written so that its instruction stream is ISSUE-bound rather than latency-bound.
//
// What the in-kernel stamps showed (tools/stamps.py, profiles/r02_stamps_*.txt): with the segments left to hipcc, the
// MFMA segment ran at 51 cycles per MFMA (every fragment read sunk to just before its MFMA: an LDS round trip per
// MFMA) and the VALU segment at 5 100 cycles for 470 instructions (register-starved: exp -> add -> rcp chains issued
// back to back, each instruction waiting for the one before).  Hence:
//   * MFMA segment  = split_mfma_asm.h (generated): one asm block, fragment reads two k-steps ahead, counted waits;
//   * VALU segment  = below: every formula is evaluated STAGE by STAGE over a batch of units (all exp2, then all adds,
//     then all rcp ...), stages pinned with sched_barrier(0), so consecutive instructions are independent and a
//     dependent one is >= 12 issue slots behind its operand.
#pragma once
#include "split_core.h"
#include "split_mfma_asm.h"

namespace rnnwf {

#define RNNWF_STAGE() __builtin_amdgcn_sched_barrier(0)

template <int NF32, int RJ, int NOUT>
struct SplitPP {
    using C = SplitCore<NF32, RJ, NOUT, 2>;
    using L = typename C::L;
    static constexpr int NT = C::NT, NU = C::NU, NUA = C::NUA, NQ = C::NQ;
    static constexpr int NB = 3 * NQ + 1;        // state quads: 3 parts x NQ k-steps + the special k-step
    // dynamic LDS of the swap pass's ping-pong kernel: the image + one checkpoint staging slot ([NU][64] f32) per wave (crnn_split_kernels.h)
    static constexpr size_t LDS_WITH_SLOTS = ((L::BYTES + 15) / 16) * 16 + 8 * (size_t)NU * 256;
    static constexpr int NP = NUA / 2;           // packed registers per part (two units each)
    using Asm = MfmaSegAsm<NT, NQ>;
    static_assert(Asm::kAvailable, "no hand-scheduled MFMA segment generated for this layout (tools/gen_split_mfma_asm.py)");
    static_assert(NP == 4 * NQ, "mode 2: the aligned units fill whole quads");

    static __device__ __forceinline__ unsigned lds_address(const void* p) {
        return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
    }

    // accumulators <- bias + one-hot input rows of input spin `sig` (table [sigma][tile][lane half][16])
    static __device__ __forceinline__ void preload(const char* lds, int sig, int lane, f32x16 (&acc)[NT]) {
        const f32x16* ci = reinterpret_cast<const f32x16*>(lds + L::OFF_CI) + (size_t)sig * NT * 2 + (lane >> 5);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = ci[t * 2];
    }

    static __device__ __forceinline__ void mfma_seg(const char* lds, const u32x4 (&B)[NB], f32x16 (&acc)[NT], int lane) {
        Asm::run(lds_address(lds + L::OFF_A) + (unsigned)lane * 16u, B, acc);
    }

    // h -> three bf16 parts, as the B quads of the MFMA segment.  Same arithmetic as SplitCore::split (every step
    // exact), evaluated stage by stage over all pairs.
    static __device__ __forceinline__ void split(const float (&h)[NU], u32x4 (&B)[NB]) {
        unsigned p1[NP], p2[NP], p3[NP];
        float r0[NP], r1[NP];
        const float x = h[NU - 1];                         // the special unit
        unsigned q1, q2, q3;
        float xr;
        RNNWF_STAGE();
#pragma unroll
        for (int i = 0; i < NP; ++i) p1[i] = cvt_pk_bf16(h[2 * i], h[2 * i + 1]);
        q1 = cvt_pk_bf16(x, x);
        RNNWF_STAGE();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            r0[i] = __uint_as_float(p1[i] << 16);
            r1[i] = __uint_as_float(p1[i] & 0xffff0000u);
        }
        xr = __uint_as_float(q1 << 16);
        RNNWF_STAGE();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            r0[i] = h[2 * i] - r0[i];
            r1[i] = h[2 * i + 1] - r1[i];
        }
        xr = x - xr;
        RNNWF_STAGE();
#pragma unroll
        for (int i = 0; i < NP; ++i) p2[i] = cvt_pk_bf16(r0[i], r1[i]);
        q2 = cvt_pk_bf16(xr, xr);
        RNNWF_STAGE();
        float s0[NP], s1[NP], xs;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            s0[i] = __uint_as_float(p2[i] << 16);
            s1[i] = __uint_as_float(p2[i] & 0xffff0000u);
        }
        xs = __uint_as_float(q2 << 16);
        RNNWF_STAGE();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            s0[i] = r0[i] - s0[i];
            s1[i] = r1[i] - s1[i];
        }
        xs = xr - xs;
        RNNWF_STAGE();
#pragma unroll
        for (int i = 0; i < NP; ++i) p3[i] = cvt_pk_bf16(s0[i], s1[i]);
        q3 = cvt_pk_bf16(xs, xs);
        RNNWF_STAGE();
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                B[q][j] = p1[4 * q + j];
                B[NQ + q][j] = p2[4 * q + j];
                B[2 * NQ + q][j] = p3[4 * q + j];
            }
        // special k-step: K order {h1, h2, h3, h1, h2, h1} (A side {w1, w1, w1, w2, w2, w3}), two entries per register
        B[3 * NQ][0] = (q1 & 0xffffu) | (q2 & 0xffff0000u);
        B[3 * NQ][1] = (q3 & 0xffffu) | (q1 & 0xffff0000u);
        B[3 * NQ][2] = (q2 & 0xffffu) | (q1 & 0xffff0000u);
        B[3 * NQ][3] = 0u;
        RNNWF_STAGE();
    }

    // accumulator slots of owned unit e: (r, u, candidate) pre-activations (e is a constant after unrolling)
    static __device__ __forceinline__ void slots(const f32x16 (&acc)[NT], int e, float& ar, float& au, float& ac) {
        if (e < 16 * NF32) {
            ar = acc[3 * (e / 16)][e % 16];
            au = acc[3 * (e / 16) + 1][e % 16];
            ac = acc[3 * (e / 16) + 2][e % 16];
        } else {
            const int j = e - 16 * NF32;
            ar = acc[3 * NF32 + (j) / 16][(j) % 16];
            au = acc[3 * NF32 + (RJ + j) / 16][(RJ + j) % 16];
            ac = acc[3 * NF32 + (2 * RJ + j) / 16][(2 * RJ + j) % 16];
        }
    }

    // head rows of the state that ENTERED the step (logits of the previous site), straight from the accumulators
    static __device__ __forceinline__ void head_lagged(const f32x16 (&acc)[NT], float (&z)[NOUT]) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) z[o] = acc[3 * NF32 + (L::HEAD_SLOT + o) / 16][(L::HEAD_SLOT + o) % 16];
    }

    // gate arithmetic of units [E0, E1), one stage at a time: r = sigmoid, u = sigmoid, c = tanh(xc + r q), h' = c + u (h - c)
    template <int E0, int E1>
    static __device__ __forceinline__ void gate_batch(const f32x16 (&acc)[NT], const float* xcp, float (&h)[NU]) {
        constexpr int n = E1 - E0;
        float ar[n], au[n], ac[n], xc[n];
#pragma unroll
        for (int j = 0; j < n; ++j) xc[j] = xcp[E0 + j];        // LDS reads: in flight during the first three stages
#pragma unroll
        for (int j = 0; j < n; ++j) slots(acc, E0 + j, ar[j], au[j], ac[j]);
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) { ar[j] = __builtin_amdgcn_exp2f(ar[j]); au[j] = __builtin_amdgcn_exp2f(au[j]); }
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) { ar[j] = 1.0f + ar[j]; au[j] = 1.0f + au[j]; }
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) { ar[j] = __builtin_amdgcn_rcpf(ar[j]); au[j] = __builtin_amdgcn_rcpf(au[j]); }
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) ac[j] = fmaf(ar[j], ac[j], xc[j]);
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) ac[j] = __builtin_amdgcn_exp2f(ac[j]);
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) ac[j] = 1.0f + ac[j];
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) ac[j] = __builtin_amdgcn_rcpf(ac[j]);
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) ac[j] = fmaf(2.0f, ac[j], -1.0f);
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) ar[j] = h[E0 + j] - ac[j];
        RNNWF_STAGE();
        // three-address form spelled out: left to itself hipcc copies c into the state's register and issues the two-address
        // v_fmac (25 v_mov per wave-step).  Plain VALU producer, VALU consumers (head, re-split): no software hazard.
#pragma unroll
        for (int j = 0; j < n; ++j) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(h[E0 + j]) : "v"(au[j]), "v"(ar[j]), "v"(ac[j]));
        RNNWF_STAGE();
    }

    static __device__ __forceinline__ void gates(const char* lds, int sig, const f32x16 (&acc)[NT], float (&h)[NU], int lane) {
        const float* xcp = reinterpret_cast<const float*>(lds + L::OFF_XC) + (size_t)((sig * 2 + (lane >> 5)) * L::NUP);
        constexpr int HALF = (NU + 1) / 2;
        gate_batch<0, HALF>(acc, xcp, h);
        gate_batch<HALF, NU>(acc, xcp, h);
    }

    // head rows on the new state: four independent partial sums per row, halves joined with v_permlane32_swap
    static __device__ __forceinline__ void head(const char* lds, const float (&h)[NU], int lane, float (&z)[NOUT]) {
        const float* wd = reinterpret_cast<const float*>(lds + L::OFF_WD) + (lane >> 5) * L::NUP * NOUT;
        float part[NOUT][4];
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int c = 0; c < 4; ++c) part[o][c] = 0.0f;
        constexpr int CH = NOUT == 1 ? NU : 8;               // units per chunk of head weights held in registers
#pragma unroll
        for (int e0 = 0; e0 < NU; e0 += CH) {
            constexpr int dummy = 0; (void)dummy;
            float w[CH * NOUT];
#pragma unroll
            for (int i = 0; i < CH * NOUT; ++i) w[i] = e0 * NOUT + i < NU * NOUT ? wd[e0 * NOUT + i] : 0.0f;
#pragma unroll
            for (int e = e0; e < e0 + CH && e < NU; ++e)
#pragma unroll
                for (int o = 0; o < NOUT; ++o) part[o][e & 3] = fmaf(h[e], w[(e - e0) * NOUT + o], part[o][e & 3]);
        }
        const float* bd = reinterpret_cast<const float*>(lds + L::OFF_BD);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float s = (part[o][0] + part[o][1]) + (part[o][2] + part[o][3]);
            const unsigned u = __float_as_uint(s);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // [0]: lower half's sum everywhere, [1]: upper half's
            z[o] = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) + bd[o];
        }
        RNNWF_STAGE();
    }
};


// ---- stacked layers: one GRU layer ABOVE the first, in the two segments of the ping-pong kernels (split_core.h: SplitUpperLayout) ----
// The layer's state travels only as its three bf16 parts (the B quads of the H block): a wave at two per SIMD has 256 registers,
// and seven accumulator tiles (112) + the fragment ring of the MFMA segment (40) + the quads of x and h (80) leave no room for a
// second, f32 copy of the state.  The gate arithmetic rebuilds h = h1 + h2 + h3 from the parts - exact, smallest first: the split
// was exact - which costs ~5 VALU instructions per unit in a segment that has room for them (190 MFMAs = 6 080 matrix-pipe cycles
// against ~4 000 issue cycles of vector work).
template <int NF32, int RJ, int NOUT>
struct SplitPPUpper {
    using P0 = SplitPP<NF32, RJ, NOUT>;
    using U = SplitUpperLayout<NF32, RJ, NOUT>;
    using L = typename U::L0;
    static constexpr int NU = P0::NU, NUA = P0::NUA, NQ = P0::NQ, NB = P0::NB, NP = P0::NP, NTA = U::NTA;
    using Asm = MfmaSegAsm<U::NTB, NQ>;
    static_assert(Asm::kAvailable, "no hand-scheduled MFMA segment generated for this layout (tools/gen_split_mfma_asm.py)");
    static_assert(RJ <= 16 && 2 * RJ >= 16 && 3 * RJ + NOUT <= 32, "upper layer: r slots in mixed tile 0, candidate slots and head rows in mixed tile 1");

    static __device__ __forceinline__ void stage(char* lds, const void* wup) {
        const uint4* src = reinterpret_cast<const uint4*>(wup);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < (int)(U::BYTES / 16); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }

    // X block (r, u, y, mixed 0, mixed 1 X), every tile from zero, then H block (r, u, mixed 0 continued; q, mixed 1 H from zero): the
    // generated block of the one-layer step, twice.  The biases ride in the special k-step (ones(): the B side of their K entries).
    static __device__ __forceinline__ void mfma_seg(const char* lds, const u32x4 (&BX)[NB], const u32x4 (&BH)[NB], f32x16 (&acc)[NTA], int lane) {
        Asm::run_tiles_from_zero(P0::lds_address(lds + U::OFF_AX) + (unsigned)lane * 16u, BX, acc[0], acc[1], acc[2], acc[4], acc[5]);
        Asm::run_tiles_zero_2_4(P0::lds_address(lds + U::OFF_AH) + (unsigned)lane * 16u, BH, acc[0], acc[1], acc[3], acc[4], acc[6]);
    }
    // h -> quads, with the bias entries' ones
    static __device__ __forceinline__ void split(const float (&h)[NU], u32x4 (&B)[NB]) {
        P0::split(h, B);
        B[3 * NQ][3] = U::B_ONES;
    }

    // head rows of the state that ENTERED the step: H block, mixed tile 1
    static __device__ __forceinline__ void head_lagged(const f32x16 (&acc)[NTA], float (&z)[NOUT]) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) z[o] = acc[6][L::HEAD_SLOT - 16 + o];
    }

    // gate arithmetic of units [E0, E1), stage by stage (see SplitPP::gate_batch): the candidate's input term comes out of the y
    // accumulators, the old state out of the quads B - unpacked and summed beside the candidate's exp / rcp chain, whose latencies
    // it fills (and only then: three more live values per unit any earlier would not fit the register file)
    template <int E0, int E1>
    static __device__ __forceinline__ void gate_batch(const f32x16 (&acc)[NTA], const u32x4 (&B)[NB], float (&h)[NU]) {
        constexpr int n = E1 - E0;
        float ar[n], au[n], aq[n], ay[n], p1[n], p2[n], p3[n];
#pragma unroll
        for (int j = 0; j < n; ++j) {
            const int e = E0 + j;
            if (e < 16 * NF32) {
                ar[j] = acc[0][e]; au[j] = acc[1][e]; ay[j] = acc[2][e]; aq[j] = acc[3][e];
            } else {
                const int r = e - 16 * NF32, su = RJ + r, sc = 2 * RJ + r;
                ar[j] = acc[4][r];
                au[j] = su < 16 ? acc[4][su] : acc[5][su - 16] + acc[6][su - 16];      // the u halves of the two blocks
                ay[j] = acc[5][sc - 16];
                aq[j] = acc[6][sc - 16];
            }
        }
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) { ar[j] = __builtin_amdgcn_exp2f(ar[j]); au[j] = __builtin_amdgcn_exp2f(au[j]); }
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) { ar[j] = 1.0f + ar[j]; au[j] = 1.0f + au[j]; }
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) { ar[j] = __builtin_amdgcn_rcpf(ar[j]); au[j] = __builtin_amdgcn_rcpf(au[j]); }
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) aq[j] = fmaf(ar[j], aq[j], ay[j]);
        RNNWF_STAGE();
        // the three parts of the old state as f32 (a bf16 is the upper half of an f32)
        auto part = [&](int e, int k) -> float {
            if (e < NUA) {
                const int i = e >> 1;
                const unsigned w = B[k * NQ + i / 4][i % 4];
                return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
            }
            // the special unit: quad 3 NQ = {h1 | h2 << 16, h3 | h1 << 16, h2 | h1 << 16, 0}
            return __uint_as_float(k == 0 ? (B[3 * NQ][0] << 16) : k == 1 ? (B[3 * NQ][0] & 0xffff0000u) : (B[3 * NQ][1] << 16));
        };
#pragma unroll
        for (int j = 0; j < n; ++j) { aq[j] = __builtin_amdgcn_exp2f(aq[j]); p3[j] = part(E0 + j, 2); p2[j] = part(E0 + j, 1); }
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) { aq[j] = 1.0f + aq[j]; p2[j] = p3[j] + p2[j]; p1[j] = part(E0 + j, 0); }
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) { aq[j] = __builtin_amdgcn_rcpf(aq[j]); p1[j] = p2[j] + p1[j]; }      // p1 = the old state, exactly
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) aq[j] = fmaf(2.0f, aq[j], -1.0f);
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) ar[j] = p1[j] - aq[j];
        RNNWF_STAGE();
#pragma unroll
        for (int j = 0; j < n; ++j) h[E0 + j] = fmaf(au[j], ar[j], aq[j]);
        RNNWF_STAGE();
    }

    // new state (f32) of this lane's units from the accumulators and the old state's quads, in two calls: the remainder units -
    // which release the three mixed tiles' registers - and the 16 units of the full tiles; the kernels request the next step's
    // input between the two, into the registers just released
    static __device__ __forceinline__ void gates_rem(const f32x16 (&acc)[NTA], const u32x4 (&B)[NB], float (&h)[NU]) {
        gate_batch<16 * NF32, NU>(acc, B, h);
    }
    static __device__ __forceinline__ void gates_full(const f32x16 (&acc)[NTA], const u32x4 (&B)[NB], float (&h)[NU]) {
        gate_batch<0, 8 * NF32>(acc, B, h);
        gate_batch<8 * NF32, 16 * NF32>(acc, B, h);
    }

    // head rows on the new state (top layer only): as SplitPP::head, on this image's tables
    static __device__ __forceinline__ void head(const char* lds, const float (&h)[NU], int lane, float (&z)[NOUT]) {
        const float* wd = reinterpret_cast<const float*>(lds + U::OFF_WD) + (lane >> 5) * L::NUP * NOUT;
        float part[NOUT][4];
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int c = 0; c < 4; ++c) part[o][c] = 0.0f;
#pragma unroll
        for (int e = 0; e < NU; ++e)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) part[o][e & 3] = fmaf(h[e], wd[e * NOUT + o], part[o][e & 3]);
        const float* bd = reinterpret_cast<const float*>(lds + U::OFF_BD);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float s = (part[o][0] + part[o][1]) + (part[o][2] + part[o][3]);
            const unsigned u = __float_as_uint(s);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            z[o] = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) + bd[o];
        }
        RNNWF_STAGE();
    }
};

}  // namespace rnnwf
