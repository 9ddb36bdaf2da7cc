// split_core.h - "bf16x3" engine of the GRU step: f32-accurate matrix products on the bf16 matrix core.
//
// Why: on gfx950 the f32-input MFMA runs at the f32 vector rate and its time ADDS to the VALU time of the gate
// arithmetic (profiles/r01_c_*), while the bf16 MFMA is 16x faster.  Every f32 operand is represented EXACTLY as
// the sum of three bf16 numbers (8 + 8 + 8 significand bits = the 24 of an f32):
//        w = w1 + w2 + w3,   h = h1 + h2 + h3                      (w split once on the host, h every site)
// and the product is taken as the six bf16 MFMA products whose magnitude is >= 2^-16 of the leading one,
//        w h ~ w1 h1 + (w2 h1 + w1 h2) + (w3 h1 + w2 h2 + w1 h3),
// accumulated in f32, smallest terms first.  bf16 x bf16 products are exact in f32; the dropped terms
// (w2 h3, w3 h2, w3 h3) are below 2^-24 of |w h| each - the size of one f32 rounding.  So the result has f32
// accuracy (tests: same tolerances as the f32-MFMA engine, both compared with the float64 oracle).
//
// Shapes: v_mfma_f32_32x32x16_bf16, 32 chains per wave (columns = lane & 31), 32-row tiles.  Lane (c, hh = lane>>5),
// accumulator register rho of a tile holds row (rho & 3) + 8 (rho >> 2) + 4 hh.  The lane therefore owns, for its
// chain, the units
//      entry e < 16 NF32           : unit 32 (e/16) + row(e % 16, hh)            (full 32-unit blocks)
//      entry 16 NF32 + j, j < RJ   : unit 32 NF32 + hh RJ + j                    (remainder units)
// and, as for the f32 engine, that is also the K-ownership of the B operand (lane half hh supplies the 8
// consecutive K entries 8x .. 8x+7 of quad x), so the new state is re-split in registers and fed straight back.
// Gate rows: full block t -> tiles 3t + {r, u, c}; remainder units -> NMIX "mixed" tiles, slot g RJ + j.
// Two extra K entries per part carry the bias (entry NU, value 1) and the one-hot input (entry NU+1, value sigma).
//
// Layout modes (MODE):
//  0  classic: every part of every product padded to whole k-steps, bias / input as two extra K entries (above).
//  1  flat K-packing (num_units <= 36): the six products are ONE chain over a concatenated K axis - a lane's 3 x NU/2
//     packed state registers, product by product, cut into k-steps of four registers without per-product padding;
//     one A fragment per (k-step, tile), packed on the host for exactly that list.
//  2  aligned + special unit (37 <= num_units <= 50, the headline width): each lane half owns 24 "aligned" units whose
//     three parts are exactly three k-steps each (no padding at all: 18 k-steps for the six products) plus ONE special
//     unit (48 + hh) whose six products {w1 h1, w1 h2, w1 h3, w2 h1, w2 h2, w3 h1} fill six of the eight K entries of
//     a 19th k-step.  95 MFMAs per 32-chain wave-step instead of 120; A fragments stay shared between products
//     (image 50 KB instead of 60 KB).
//  3  the same K-packing at 69 <= num_units <= 100 (config 5): 48 aligned units per lane half (6 k-steps per part) + TWO
//     special units whose 12 products fill two extra k-steps: 38 k-steps x 10 tiles = 380 MFMAs (mode 0 would need 420).
//     Its 200 KB of fragments exceed LDS: the regular w3 fragments stay in global memory (STREAM, step_stream below).
// Modes 1 - 3 take bias and one-hot input off the K axis: they are the accumulators' initial value, read from a
// table [sigma][tile][lane half][16] (rows differ, chains do not).  Modes 2 and 3 also carry the head rows in spare
// slots of the mixed tiles (pack_split.h).
#pragma once
#include <type_traits>

#include "gru_core.h"

namespace rnnwf {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NF32_, int RJ_, int NOUT_ = 1, int MODE_ = 0>
struct SplitLayout {
    static constexpr int NOUT = NOUT_;                  // head rows: 1 (pRNN logit difference), 3 (cRNN: + 2 phase logits)
    static constexpr int NF32 = NF32_, RJ = RJ_, MODE = MODE_;
    static constexpr int NMIX = (3 * RJ + 15) / 16;
    static constexpr int NT = 3 * NF32 + NMIX;          // 32-row output tiles
    static constexpr int NU = 16 * NF32 + RJ;           // hidden units owned by one lane (mode 2: the last one is special)
    static constexpr int NUA = MODE == 2 ? NU - 1 : MODE == 3 ? (NU / 8) * 8 : NU;   // units that travel in the regular packed registers
    static constexpr int NS = NU - NUA;                 // special units (modes 2, 3): their six products fill K entries of extra k-steps
    static constexpr int KSP = MODE == 2 ? 1 : MODE == 3 ? (6 * NS + 7) / 8 : 0;     // special k-steps
    static constexpr int NE = MODE == 0 ? NU + 2 : NUA; // mode 0: + bias entry + input entry
    static constexpr int NQ = (NE + 7) / 8;             // bf16x8 quads per part (modes 0, 2, 3)
    static constexpr int NRM = MODE == 1 ? NU / 2 : 4 * NQ;     // packed 32-bit registers per part
    static constexpr int NR = MODE == 2 ? NRM + 1 : NRM;        // declared per part: mode 2 keeps the special k-step's B registers in slot NRM
    static constexpr int KS = MODE == 1 ? (6 * NRM + 3) / 4 : 6 * NQ + KSP;   // k-steps (MFMAs per tile)
    static constexpr int NUP = ((NU + 3) / 4) * 4;
    static constexpr int HP = 32 * NF32 + 2 * RJ;       // padded hidden size
    static constexpr int HEAD_SLOT = 3 * RJ;            // mode 2 / STREAM: mixed-tile slots [3 RJ, 3 RJ + NOUT) hold the head rows (pack_split.h)
    static_assert(MODE != 2 || 3 * RJ + NOUT <= 16 * NMIX, "mode 2: the head rows need spare slots in the mixed tiles");
    static_assert((MODE != 2 && MODE != 3) || (NUA % 8 == 0), "modes 2, 3: the aligned units fill whole k-steps");
    static_assert(MODE != 3 || (NS >= 1 && NUA == 16 * NF32), "mode 3: the remainder units are the special ones");
    static_assert(MODE != 1 || (NU % 2 == 0), "mode 1: units are packed in pairs");
    // A image: mode 0 [NT][3][NQ][64] x 16 B; mode 1 [KS][NT][64] x 16 B; mode 2 [NT][3][NQ][64] then [NT][64] (special)
    static constexpr size_t OFF_A = 0;
    static constexpr size_t SZ_A = MODE == 1 ? (size_t)KS * NT * 64 * 16 : (size_t)NT * (3 * NQ + KSP) * 64 * 16;
    static constexpr size_t OFF_ASP = (size_t)NT * 3 * NQ * 64 * 16;           // modes 2, 3: special fragments [NT][KSP][64] x 16 B
    static constexpr size_t OFF_CI = OFF_A + SZ_A;                             // modes 1, 2: [2 sigma][NT][2 hh][16] f32
    static constexpr size_t OFF_XC = OFF_CI + (MODE != 0 ? (size_t)2 * NT * 2 * 16 * 4 : 0);   // [2 sigma][2 hh][NUP] f32 (scaled)
    static constexpr size_t OFF_WD = OFF_XC + (size_t)2 * 2 * NUP * 4;         // [2 hh][NUP][NOUT] f32 head weights
    static constexpr size_t OFF_BD = OFF_WD + (size_t)2 * NUP * NOUT * 4;      // [4] f32 head biases (padded)
    static constexpr size_t BYTES = OFF_BD + 16;
    // Mode 3 (69..100 units; 200 KB of fragments for 160 KB of LDS): the regular w3 fragments - used by ONE of the six
    // products - stay in global memory and are read through L2 (STREAM); LDS holds [NT][2][NQ] regular fragments, then
    // everything from OFF_ASP on (special fragments, tables) LSHIFT bytes lower than in the global image.
    static constexpr bool RIDERS = MODE == 3;                                  // kernels: SplitCore::step_stream (one wave per SIMD)
    static constexpr bool STREAM = MODE == 3 && BYTES > 160 * 1024;            // (53..68 units: the whole mode-3 image fits LDS)
    static constexpr size_t LDS_REG = (size_t)NT * 2 * NQ * 64 * 16;           // STREAM: regular fragments of parts 0, 1
    static constexpr size_t LSHIFT = STREAM ? OFF_ASP - LDS_REG : 0;
    static constexpr size_t LDS_BYTES = BYTES - LSHIFT;
    static_assert(MODE == 3 || BYTES <= 160 * 1024, "the image must fit LDS (mode 3 streams one weight part)");
    static_assert(!STREAM || LDS_BYTES <= 160 * 1024, "mode 3: the resident part of the image must fit LDS");
    // unit owned by entry e of lane half hh
    static constexpr int unit_of(int e, int hh) {
        if (e < 16 * NF32) return 32 * (e / 16) + ((e % 16) & 3) + 8 * ((e % 16) >> 2) + 4 * hh;
        if (MODE == 2) return e < NU - 1 ? 32 * NF32 + hh * (RJ - 1) + (e - 16 * NF32) : 32 * NF32 + 2 * (RJ - 1) + hh;
        return 32 * NF32 + hh * RJ + (e - 16 * NF32);
    }
};

// Stacked GRU layers above the first on the bf16x3 engine (tf.nn.rnn_cell.MultiRNNCell, 1DTFIM/RNNwavefunction.py:32; the complex
// wave function's default is two layers, J1J2/ComplexRNNwavefunction.py:16,40), K-packed layout MODE 2 (37..50 units).
// The input x of layer l is the new state of layer l - 1: an h-wide f32 vector instead of a one-hot, so a step is TWO blocks of
// products, each shaped exactly like the first layer's image (SplitLayout<.., 2>: NT = 5 tiles, 19 k-steps, 50 KB):
//     X block (K over x):  rows r, u, y = x Wci + bci         H block (K over h):  rows r, u, q = h Wch + bch
// into SEVEN accumulator tiles
//     0: r   1: u   2: y   3: q        (units 0..31; r and u take both blocks)
//     4: mixed tile 0 = r[0..8], u[0..6] of the remainder units (both blocks)
//     5: mixed tile 1 of the X block = u[7..8] (x part), y[0..8]
//     6: mixed tile 1 of the H block = u[7..8] (h part), q[0..8], head rows of the state that entered the step
// (the two u halves of remainder units 7, 8 meet in one VALU add each).  c = tanh(y + r q).  190 MFMAs per 32-chain wave-step.
// Biases: the special k-step of MODE 2 uses six of its eight K entries per lane half; entries 6, 7 of both halves carry the constant
// 1.0 on the B side and the three bf16 parts of the row's bias on the A side (X block: b_r, b_u, b_ci; H block: b_ch, head biases),
// so every accumulator starts from the MFMA's inline zero: no preload, no table, no registers held across the VALU segment.
// One layer per kernel: the layers of a stack run as a PIPELINE of kernels over the same flip / swap tiles, layer l - 1 writing its
// new state of every step to a record buffer in HBM that layer l reads one step ahead of its use (split_kernels.h) - the images
// of two such layers (50 + 100 KB) would fill LDS, and a wave at two per SIMD cannot hold two layers' accumulators and states.
template <int NF32_, int RJ_, int NOUT_ = 1>
struct SplitUpperLayout {
    using L0 = SplitLayout<NF32_, RJ_, NOUT_, 2>;
    static_assert(NF32_ == 1 && L0::NMIX == 2, "upper-layer layout: one 32-unit block + two mixed tiles (37..50 units)");
    static constexpr int NOUT = NOUT_, NF32 = NF32_, RJ = RJ_;
    static constexpr int NTB = L0::NT;                          // tiles per block
    static constexpr int NTA = 7;                               // accumulator tiles
    static constexpr int NU = L0::NU, NUP = L0::NUP;
    static constexpr size_t SZ_BLOCK = L0::SZ_A;                // one block's A fragments, laid out as SplitLayout<.., 2>'s
    static constexpr size_t OFF_AX = 0;
    static constexpr size_t OFF_AH = SZ_BLOCK;
    static constexpr size_t OFF_WD = 2 * SZ_BLOCK;              // [2 hh][NUP][NOUT] f32 head weights (top layer)
    static constexpr size_t OFF_BD = OFF_WD + (size_t)2 * NUP * NOUT * 4;   // [4] f32 head biases (padded)
    static constexpr size_t BYTES = OFF_BD + 16;
    static_assert(BYTES <= 160 * 1024, "the upper-layer image must fit LDS");
    // accumulator tile of block tile t
    static constexpr int acc_of(bool xblock, int t) { return t < 2 ? t : t == 2 ? (xblock ? 2 : 3) : t == 3 ? 4 : (xblock ? 5 : 6); }
    // one wave-step's state record: [NG groups of 4 entries][64 lanes][4] f32 - a lane's entries 4 g .. 4 g + 3 are 16 contiguous bytes,
    // a group 1 KB - then the NU % 4 last entries as [entry][64 lanes] f32: written with dwordx4 (dword) stores, read by LDS-DMA into a
    // lane-linear staging slot (split_kernels.h).  NU floats per lane, nothing padded: the records are the pipeline's HBM traffic.
    static constexpr int NG = NU / 4, NTAIL = NU % 4;
    static constexpr int RECORD_FLOATS = NU * 64;
    static constexpr size_t SLOT_BYTES = (size_t)RECORD_FLOATS * 4;       // per-wave LDS staging slot (also holds a checkpoint: [NU][64] f32)
    static constexpr size_t LDS_BYTES = ((BYTES + 15) / 16) * 16 + 8 * SLOT_BYTES;
    static_assert(LDS_BYTES + 64 <= 160 * 1024, "image + eight staging slots must fit LDS");
    static constexpr unsigned B_ONES = 0x3F803F80u;             // register 3 of the special quad: K entries 6, 7 = bf16 1.0
};

// x -> packed (bf16(x0), bf16(x1)) with round-to-nearest-even: ONE v_cvt_pk_bf16_f32, emitted by the compiler from the
// vector conversion.  It must NOT be inline asm: hipcc's hazard recognizer does not count an asm block as a VALU
// instruction, so an MFMA reading the packed register right behind it got no wait states and multiplied STALE
// operands on some lanes (found in round 2: with a different instruction order the 20-unit cRNN swap kernel returned
// 53 distinct values for 64 copies of one configuration; tools/dbg_crnn_multipass.py).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

template <int NF32, int RJ, int NOUT = 1, int MODE = 0>
struct SplitCore {
    using L = SplitLayout<NF32, RJ, NOUT, MODE>;
    static constexpr int NT = L::NT, NU = L::NU, NUA = L::NUA, NQ = L::NQ, NR = L::NR, NRM = L::NRM, KS = L::KS;

    static __device__ __forceinline__ void stage(char* lds, const void* wimg) {
        const uint4* src = reinterpret_cast<const uint4*>(wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        if constexpr (L::STREAM) {
            // fragments of parts 0, 1 compacted to [NT][2][NQ][64], then the tables
            constexpr int PER = NQ * 64;                           // uint4 per (tile, part)
            for (int i = threadIdx.x; i < NT * 2 * PER; i += blockDim.x) {
                const int tp = i / PER, r = i - tp * PER;
                dst[i] = src[((tp >> 1) * 3 + (tp & 1)) * PER + r];
            }
            for (int i = threadIdx.x; i < (int)((L::BYTES - L::OFF_ASP) / 16); i += blockDim.x)
                dst[L::LDS_REG / 16 + i] = src[L::OFF_ASP / 16 + i];
        } else {
            for (int i = threadIdx.x; i < (int)(L::BYTES / 16); i += blockDim.x) dst[i] = src[i];
        }
        __syncthreads();
    }

    // h (this lane's NU units) -> three bf16 parts packed two per register; entries NU (bias, 1.0) and NU+1 (input
    // spin) are appended to part 1.  Every step of the split is exact: r = x - bf16(x) is representable.
    static __device__ __forceinline__ void split(const float (&h)[NU], int sig, unsigned (&R)[3][NR]) {
        float v[2 * NRM];
#pragma unroll
        for (int e = 0; e < 2 * NRM; ++e) v[e] = e < NUA ? h[e] : 0.0f;
#pragma unroll
        for (int i = 0; i < NRM; ++i) {
            const unsigned p1 = cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
            const float r0 = v[2 * i] - __uint_as_float(p1 << 16);
            const float r1 = v[2 * i + 1] - __uint_as_float(p1 & 0xffff0000u);
            const unsigned p2 = cvt_pk_bf16(r0, r1);
            const float s0 = r0 - __uint_as_float(p2 << 16);
            const float s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
            R[0][i] = p1;
            R[1][i] = p2;
            R[2][i] = cvt_pk_bf16(s0, s1);
        }
        if constexpr (MODE == 0) {
            // bias / input entries (bf16 1.0 = 0x3F80)
            constexpr int eb = NU, es = NU + 1;
            const unsigned one = 0x3F80u, sg = sig ? 0x3F80u : 0u;
            R[0][eb / 2] |= (eb & 1) ? (one << 16) : one;
            R[0][es / 2] |= (es & 1) ? (sg << 16) : sg;
        }
        if constexpr (MODE == 2) {
            // the special unit: its parts, duplicated into both halves of a register, then the B registers of the
            // special k-step in the K order {h1, h2, h3, h1, h2, h1} (A side: {w1, w1, w1, w2, w2, w3}), 2 entries each
            const float x = h[NU - 1];
            const unsigned q1 = cvt_pk_bf16(x, x);
            const float r = x - __uint_as_float(q1 << 16);
            const unsigned q2 = cvt_pk_bf16(r, r);
            const float t = r - __uint_as_float(q2 << 16);
            const unsigned q3 = cvt_pk_bf16(t, t);
            R[0][NRM] = (q1 & 0xffffu) | (q2 & 0xffff0000u);      // h1 | h2 << 16
            R[1][NRM] = (q3 & 0xffffu) | (q1 & 0xffff0000u);      // h3 | h1 << 16
            R[2][NRM] = (q2 & 0xffffu) | (q1 & 0xffff0000u);      // h2 | h1 << 16
        }
    }

    // One GRU step for 32 chains: R (split old state + bias/input entries) and hold (old state) in, new state out.
    static __device__ __forceinline__ void step(const char* lds, int sig, const unsigned (&R)[3][NR], float (&h)[NU], int lane) {
        const int hh = lane >> 5;
        asm volatile("" ::: "memory");
        f32x16 acc[NT];
        f32x16 zero;
#pragma unroll
        for (int k = 0; k < 16; ++k) zero[k] = 0.0f;       // folds into the MFMA's inline-constant C operand
        const u32x4* av = reinterpret_cast<const u32x4*>(lds + L::OFF_A) + lane;
        const float* xc = reinterpret_cast<const float*>(lds + L::OFF_XC - L::LSHIFT) + (size_t)((sig * 2 + hh) * L::NUP);
        // gate arithmetic of one owned unit (entry e) from its three accumulator slots
        auto gate = [&](int e) {
            float ar, au, ac;
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
            const float rg = Act<float>::sigmoid_scaled(ar);
            const float ug = Act<float>::sigmoid_scaled(au);
            const float cc = Act<float>::tanh_scaled(xc[e] + rg * ac);
            h[e] = cc + ug * (h[e] - cc);
        };
        // (A pair-wise variant of `gate` on v_pk_add_f32 / v_pk_fma_f32 cut the VALU instructions of a wave-step from
        //  520 to 459 and changed nothing: flip kernel 3.16 vs 3.15 ms at config 2, more s_nop padding - not kept.)
        // (weight part, state part), smallest products first.  The A fragments of k-step k+1 are read from LDS while
        // k-step k's MFMAs run (two register sets); the compiler barrier keeps later reads from being hoisted too
        // (120 fragment quads in flight would cost all the occupancy).  Two passes over the k-steps: first the mixed
        // tiles (remainder units), then the full tiles - so that the remainder units' gate arithmetic (VALU) can sit
        // between the second pass's MFMAs: the bf16 matrix pipe lets ~2 VALU ops per MFMA issue for free (measured,
        // profiles/r01_c_microbench_*), unlike the f32-input MFMA.
        constexpr int ORD[6][2] = {{2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};
        constexpr int NK = KS;
        constexpr int TF = 3 * NF32;                       // full tiles [0, TF), mixed tiles [TF, NT)
        // modes 1, 2: bias + one-hot input rows are the accumulators' initial value (loaded per pass: short live ranges)
        const f32x16* ci = reinterpret_cast<const f32x16*>(lds + L::OFF_CI - L::LSHIFT) + (size_t)sig * NT * 2 + hh;
        // fragment address and B registers of k-step k
        auto a_index = [&](int t, int k) {
            if constexpr (MODE == 1) return (k * NT + t) * 64;
            else return k < 6 * NQ ? ((t * 3 + ORD[k / NQ][0]) * NQ + k % NQ) * 64 : (NT * 3 * NQ + t) * 64;   // mode 2: special
        };
        auto b_reg = [&](int k, int j) -> unsigned {
            if constexpr (MODE == 1) {
                const int f = 4 * k + j;                   // flat register list: product by product, NRM registers each
                return f < 6 * NRM ? R[ORD[f / NRM][1]][f % NRM] : 0u;
            } else {
                if constexpr (MODE == 2) {
                    if (k >= 6 * NQ) return j < 3 ? R[j][NRM] : 0u;     // the special k-step
                }
                return R[ORD[k / NQ][1]][4 * (k % NQ) + j];
            }
        };
        auto pass = [&](auto lo_c, auto hi_c, auto fill) {
            constexpr int T0 = decltype(lo_c)::value, T1 = decltype(hi_c)::value;
            if constexpr (T1 > T0) {
                u32x4 cur[T1 - T0], nxt[T1 - T0];
                if constexpr (MODE != 0) {
#pragma unroll
                    for (int t = T0; t < T1; ++t) acc[t] = ci[t * 2];
                }
#pragma unroll
                for (int t = T0; t < T1; ++t) cur[t - T0] = av[a_index(t, 0)];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    if (k + 1 < NK) {
#pragma unroll
                        for (int t = T0; t < T1; ++t) nxt[t - T0] = av[a_index(t, k + 1)];
                    }
                    const u32x4 bq = {b_reg(k, 0), b_reg(k, 1), b_reg(k, 2), b_reg(k, 3)};
                    const bf16x8 b = __builtin_bit_cast(bf16x8, bq);
#pragma unroll
                    for (int t = T0; t < T1; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[t - T0]), b,
                                                                         (MODE == 0 && k == 0) ? zero : acc[t], 0, 0, 0);
                    fill(k);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int t = T0; t < T1; ++t) cur[t - T0] = nxt[t - T0];
                }
            }
        };
        pass(std::integral_constant<int, TF>{}, std::integral_constant<int, NT>{}, [](int) {});
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, TF>{}, [&](int k) {
            // remainder unit j after k-step 2j+1 of the full-tile pass
            if constexpr (TF > 0 && NT > TF) {
                if ((k & 1) && (k >> 1) < RJ) gate(16 * NF32 + (k >> 1));
            }
        });
        if constexpr (TF > 0 && NT > TF) {
#pragma unroll
            for (int j = NK / 2; j < RJ; ++j) gate(16 * NF32 + j);       // (only if RJ exceeds the k-steps available)
#pragma unroll
            for (int e = 0; e < 16 * NF32; ++e) gate(e);
        } else {
#pragma unroll
            for (int e = 0; e < NU; ++e) gate(e);
        }
    }

    // ---- mode 3 (53..100 units), the "riders" form of the step; above 68 units with the regular w3 fragments read through L2 ----
    // One wave per SIMD (accumulators in AGPRs, ~256 VGPRs), so a bf16 MFMA leaves room for ~5 VALU instructions of its
    // OWN wave: the step is software-pipelined around its NF32 + 1 tile passes (NF32 = 2: 53..68 units, NF32 = 3: 69..100)
    //     split stage 1 (h1)  |  pass b = 0 .. NF32-1: unit block b (3 tiles)  <- rides: split stages 2, 3 (b = 0), gates of block b-1
    //                         |  last pass: remainder units (1 tile)          <- rides: gates of block NF32-1
    //     gates of the remainder units; the head rows of the ENTERING state come out of the remainder tile (zlag).
    // The riders are laid out stage by stage - one stage of four or eight units (or two state pairs) per k-step - so that
    // neighbouring VALU instructions are independent: a wave issues in order, and a dependent exp -> add -> rcp chain
    // between two MFMAs would hold the next MFMA back for its whole latency.
    // K-step order inside a pass (6 NQ + KSP positions): products that need h1 only come first ((w2,h1), (w1,h1)), then
    // (w2,h2), (w1,h2), then (w1,h3); the NQ k-steps of (w3,h1) - whose fragments come from global memory - are spread
    // over the pass, one in front of every five LDS-fed k-steps, requested two streamed k-steps (~1 100 cycles) ahead; the
    // last two requests of a pass fetch the first two sets of the NEXT pass (`sf` carries pass 1's from site to site);
    // the special k-steps (the remainder units' 6 products each, K-packed) close the pass.  38 k-steps x 10 tiles = 380
    // MFMAs per wave-step at 100 units (the classic layout - every part padded to 7 k-steps - had 420).  At 53..68 units
    // the whole image is in LDS: the six products are all LDS-fed ((w3,h1) first), 26 k-steps x 7 tiles = 182 MFMAs.
    static constexpr int SFN = 3;                                              // tiles per pass (a unit block's r, u, c tiles)
    // Buffer loads: one resource descriptor (4 SGPRs) for the image, the lane's 16-byte slot as the VGPR offset, the
    // fragment's position as the scalar offset - flat global loads made hipcc keep 70 loop-invariant 64-bit addresses
    // (140 VGPRs) and spill them.
    typedef __amdgpu_buffer_rsrc_t StreamSrc;
    static __device__ __forceinline__ StreamSrc stream_source(const void* gimg) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(gimg) + L::OFF_A), 0, (int)L::OFF_ASP, 0x00020000);
    }
    template <int T0, int T1>
    static __device__ __forceinline__ void stream_request(StreamSrc gimg, int k, u32x4 (&sf)[SFN], int lane) {   // one set
#pragma unroll
        for (int t = T0; t < T1; ++t)
            sf[t - T0] = __builtin_amdgcn_raw_buffer_load_b128(gimg, lane * 16, ((t * 3 + 2) * NQ + k) * 64 * 16, 0);
    }
    static __device__ __forceinline__ void stream_first(StreamSrc gimg, u32x4 (&sf)[2][SFN], int lane) {
        stream_request<0, 3>(gimg, 0, sf[0], lane);
        stream_request<0, 3>(gimg, 1, sf[1], lane);
    }
    // zlag: head rows of the state that ENTERED the step (the previous site's logits), from spare slots of the mixed tile
    static __device__ __forceinline__ void step_stream(const char* lds, StreamSrc gimg, int sig, float (&h)[NU], u32x4 (&sf)[2][SFN], int lane,
                                                       float (&zlag)[NOUT]) {
        static_assert(MODE == 3 && (NF32 == 2 || NF32 == 3) && NT == 3 * NF32 + 1 && L::KSP <= 2 && (!L::STREAM || NQ % 2 == 0),
                      "step_stream: mode 3, two or three unit blocks + one mixed tile; streamed: even number of streamed k-steps");
        constexpr int KSP = L::KSP, NS = L::NS;
        constexpr bool STREAM = L::STREAM;
        const int hh = lane >> 5;
        asm volatile("" ::: "memory");
        // fragment addresses: three opaque LDS base addresses 64 KB apart + the DS instruction's 16-bit immediate (left
        // alone, hipcc keeps one address register per fragment - 350 of them, parked in AGPRs and fetched back with a
        // v_accvgpr_read per ds_read).  Regular (tile, part, quad) at ((t * 2 + part) * NQ + q) KB, special (tile, k) behind.
        typedef const __attribute__((address_space(3))) char* LdsPtr;
        LdsPtr abase[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            abase[b] = (LdsPtr)(lds + lane * 16 + b * 65536);
            asm volatile("" : "+v"(abase[b]));
        }
        const float* xc = reinterpret_cast<const float*>(lds + L::OFF_XC - L::LSHIFT) + (size_t)((sig * 2 + hh) * L::NUP);
        const f32x16* ci = reinterpret_cast<const f32x16*>(lds + L::OFF_CI - L::LSHIFT) + (size_t)sig * NT * 2 + hh;
        // LDS-fed products (weight part, state part) in the order the state parts become available; where the whole image is in
        // LDS (53..68 units) the product (w3, h1) is the first of them, otherwise it is the streamed one
        constexpr int NPL = STREAM ? 5 : 6;                         // LDS-fed products
        constexpr int ORD[6][2] = {{STREAM ? 1 : 2, 0}, {STREAM ? 0 : 1, 0}, {STREAM ? 1 : 0, STREAM ? 1 : 0},
                                   {STREAM ? 0 : 1, 1}, {0, STREAM ? 2 : 1}, {0, 2}};
        constexpr int NLR = NPL * NQ;                               // LDS-fed regular k-steps; k-steps NLR .. NLR + KSP - 1: special
        constexpr int NL = NLR + KSP;
        constexpr int PARTS = STREAM ? 2 : 3;                       // weight parts resident in LDS
        auto frag = [&](int t, int kl) -> u32x4 {
            const int off = kl < NLR ? ((t * PARTS + ORD[kl / NQ][0]) * NQ + kl % NQ) * 1024
                                     : (int)(L::OFF_ASP - L::LSHIFT) + (t * KSP + (kl - NLR)) * 1024;
            return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(abase[off >> 16] + (off & 0xffff));
        };
        // ---- the split, pair by pair (entries 2 i, 2 i + 1 -> register i of every part); every step is exact
        constexpr int NPR = NUA / 2;                                // pairs of aligned units
        unsigned R[3][NRM];
        unsigned RS[KSP][4];                                        // B registers of the special k-steps
        float res0[NPR], res1[NPR];
#pragma unroll
        for (int i = 0; i < NRM; ++i) {
            R[0][i] = i < NPR ? cvt_pk_bf16(h[2 * i], h[2 * i + 1]) : 0u;
            R[1][i] = 0u;
            R[2][i] = 0u;
        }
        auto split2 = [&](int i) {                                  // h2 of pair i
            if (i < NPR) {
                res0[i] = h[2 * i] - __uint_as_float(R[0][i] << 16);
                res1[i] = h[2 * i + 1] - __uint_as_float(R[0][i] & 0xffff0000u);
                R[1][i] = cvt_pk_bf16(res0[i], res1[i]);
            }
        };
        auto split3 = [&](int i) {                                  // h3 of pair i
            if (i < NPR) {
                const float s0 = res0[i] - __uint_as_float(R[1][i] << 16);
                const float s1 = res1[i] - __uint_as_float(R[1][i] & 0xffff0000u);
                R[2][i] = cvt_pk_bf16(s0, s1);
            }
        };
        // special units: parts duplicated into both halves of a register, then entry 6 s + i = part {h1,h2,h3,h1,h2,h1}[i]
        auto split_special = [&]() {
            unsigned q[NS][3];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
                const float x = h[NUA + sp];
                q[sp][0] = cvt_pk_bf16(x, x);
                const float r = x - __uint_as_float(q[sp][0] << 16);
                q[sp][1] = cvt_pk_bf16(r, r);
                const float t = r - __uint_as_float(q[sp][1] << 16);
                q[sp][2] = cvt_pk_bf16(t, t);
            }
            constexpr int PH_[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int k = 0; k < KSP; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e0 = 8 * k + 2 * j, e1 = e0 + 1;
                    const unsigned lo = e0 < 6 * NS ? (q[e0 / 6][PH_[e0 % 6]] & 0xffffu) : 0u;
                    const unsigned hi = e1 < 6 * NS ? (q[e1 / 6][PH_[e1 % 6]] & 0xffff0000u) : 0u;
                    RS[k][j] = lo | hi;
                }
        };
        auto slots = [&](int e, const f32x16* acc, int T0, float& ar, float& au, float& ac) {     // acc: the pass's tiles, tile t at acc[t - T0]
            if (e < 16 * NF32) {
                ar = acc[3 * (e / 16) - T0][e % 16];
                au = acc[3 * (e / 16) + 1 - T0][e % 16];
                ac = acc[3 * (e / 16) + 2 - T0][e % 16];
            } else {
                const int j = e - 16 * NF32;
                ar = acc[3 * NF32 + (j) / 16 - T0][(j) % 16];
                au = acc[3 * NF32 + (RJ + j) / 16 - T0][(RJ + j) % 16];
                ac = acc[3 * NF32 + (2 * RJ + j) / 16 - T0][(2 * RJ + j) % 16];
            }
        };
        // gate arithmetic of four units (entries e0 .. e0 + 3) in nine stages: r = sigmoid, u = sigmoid, c = tanh(xc + r q),
        // h' = c + u (h - c) on the pre-scaled accumulators (Act<float>)
        constexpr int NPOS = (STREAM ? 6 : 6) * NQ + KSP;           // k-step positions of a pass
        constexpr int GB = NPOS >= 36 ? 4 : 8;                      // units per rider batch: 9 stages x 16 / GB batches must fit a pass
        static_assert(9 * (16 / GB) <= NPOS, "the gate riders of a unit block must fit the positions of a pass");
        float gr[GB], gu[GB], gc[GB], gx[GB];
        auto gate_stage = [&](int e0, int st, const f32x16* acc, int T0) {
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int e = e0 + u;
                if (e >= NU) continue;
                switch (st) {
                    case 0: slots(e, acc, T0, gr[u], gu[u], gc[u]); gx[u] = xc[e]; break;
                    case 1: gr[u] = __builtin_amdgcn_exp2f(gr[u]); gu[u] = __builtin_amdgcn_exp2f(gu[u]); break;
                    case 2: gr[u] = 1.0f + gr[u]; gu[u] = 1.0f + gu[u]; break;
                    case 3: gr[u] = __builtin_amdgcn_rcpf(gr[u]); gu[u] = __builtin_amdgcn_rcpf(gu[u]); break;
                    case 4: gc[u] = fmaf(gr[u], gc[u], gx[u]); break;
                    case 5: gc[u] = __builtin_amdgcn_exp2f(gc[u]); break;
                    case 6: gc[u] = 1.0f + gc[u]; break;
                    case 7: gc[u] = __builtin_amdgcn_rcpf(gc[u]); break;
                    case 8: gc[u] = fmaf(2.0f, gc[u], -1.0f); h[e] = fmaf(gu[u], h[e] - gc[u], gc[u]); break;
                }
            }
        };
        // the gates of unit block b (16 units, accumulators `acc` of the pass [T0, T0 + 3)) as riders of a later pass
        auto ride_block = [&](int b, int pos, const f32x16* acc, int T0) {
            if (pos < 9 * (16 / GB)) gate_stage(16 * b + GB * (pos / 9), pos % 9, acc, T0);
        };
        // One pass = the 6 NQ + KSP k-steps of tiles [T0, T1) in the position order  S L L L L L  S L L L L L ... P P
        // (S: a streamed k-step of (w3, h1), L: LDS-fed, P: special).  Fragments are requested ahead of their use: LDS-fed
        // sets LA k-steps ahead (a three-tile k-step is only ~96 cycles of matrix pipe), streamed sets two S k-steps ahead.
        auto pass = [&](auto t0_c, auto t1_c, auto nt0_c, auto nt1_c, f32x16* acc, auto fill) {
            constexpr int T0 = decltype(t0_c)::value, T1 = decltype(t1_c)::value, NTP = T1 - T0;
            constexpr int N0 = decltype(nt0_c)::value, N1 = decltype(nt1_c)::value;       // tiles of the pass that follows
            constexpr int LA = 2;                     // LDS look-ahead in k-steps (measured 1 / 2 / 3 / 4: 204.5 / 199.9 / 200.8 / 202.3 ms at config 5)
            auto mfma_all = [&](const u32x4* fr, const unsigned* Rp, int q) {
                const u32x4 bq = {Rp[4 * q], Rp[4 * q + 1], Rp[4 * q + 2], Rp[4 * q + 3]};
                const bf16x8 b = __builtin_bit_cast(bf16x8, bq);
#pragma unroll
                for (int t = 0; t < NTP; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[t]), b, acc[t], 0, 0, 0);
            };
#pragma unroll
            for (int t = 0; t < NTP; ++t) acc[t] = ci[(T0 + t) * 2];    // bias + one-hot input rows (+ head biases)
            u32x4 ring[LA + 1][NTP];
#pragma unroll
            for (int a = 0; a < LA; ++a)
#pragma unroll
                for (int t = 0; t < NTP; ++t) ring[a][t] = frag(T0 + t, a);
            auto lds_kstep = [&](int kl, int pos) {
                if (kl + LA < NL) {
#pragma unroll
                    for (int t = 0; t < NTP; ++t) ring[(kl + LA) % (LA + 1)][t] = frag(T0 + t, kl + LA);
                }
                if (kl < NLR) mfma_all(ring[kl % (LA + 1)], R[ORD[kl / NQ][1]], kl % NQ);
                else mfma_all(ring[kl % (LA + 1)], RS[kl - NLR], 0);
                fill(pos);
                asm volatile("" ::: "memory");
            };
            if constexpr (STREAM) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    mfma_all(sf[j & 1], R[0], j);                   // streamed k-step j: w3 x h1
                    fill(6 * j);
                    if (j + 2 < NQ) stream_request<T0, T1>(gimg, j + 2, sf[j & 1], lane);
                    else stream_request<N0, N1>(gimg, j + 2 - NQ, sf[j & 1], lane);    // sets 0, 1 of the next pass (or of the next site)
#pragma unroll
                    for (int i = 0; i < 5; ++i) lds_kstep(5 * j + i, 6 * j + 1 + i);
                }
#pragma unroll
                for (int k = 0; k < KSP; ++k) lds_kstep(NLR + k, 6 * NQ + k);
            } else {
#pragma unroll
                for (int kl = 0; kl < NL; ++kl) lds_kstep(kl, kl);
            }
        };
        using T0c = std::integral_constant<int, 0>;
        using T3c = std::integral_constant<int, 3>;
        using T6c = std::integral_constant<int, 6>;
        using T9c = std::integral_constant<int, 9>;
        using TNc = std::integral_constant<int, NT>;
        // pass 1: the split's stage 2 rides positions [0, 2 NQ), stage 3 [2 NQ, 4 NQ), the special registers position 4 NQ: h2 is
        // first needed by the third (streamed layout) / fourth (resident layout) LDS-fed product, h3 by the last one
        constexpr int S2 = (NPR + 2 * NQ - 1) / (2 * NQ);           // pairs per position, a stage done within 2 NQ positions
        f32x16 accA[3], accB[3];
        auto split_riders = [&](int pos) {
#pragma unroll
            for (int u = 0; u < S2; ++u) {
                if (pos < 2 * NQ) split2(pos * S2 + u);
                else if (pos < 4 * NQ) split3((pos - 2 * NQ) * S2 + u);
            }
            if (pos == 4 * NQ) split_special();
        };
        f32x16* acc_last;                                           // the remainder tile's accumulators
        if constexpr (NF32 == 3) {
            pass(T0c{}, T3c{}, T3c{}, T6c{}, accA, split_riders);
            pass(T3c{}, T6c{}, T6c{}, T9c{}, accB, [&](int pos) { ride_block(0, pos, accA, 0); });
            pass(T6c{}, T9c{}, T9c{}, TNc{}, accA, [&](int pos) { ride_block(1, pos, accB, 3); });
            pass(T9c{}, TNc{}, T0c{}, T3c{}, accB, [&](int pos) { ride_block(2, pos, accA, 6); });
            acc_last = accB;
        } else {
            pass(T0c{}, T3c{}, T3c{}, T6c{}, accA, split_riders);
            pass(T3c{}, T6c{}, T6c{}, TNc{}, accB, [&](int pos) { ride_block(0, pos, accA, 0); });
            pass(T6c{}, TNc{}, T0c{}, T3c{}, accA, [&](int pos) { ride_block(1, pos, accB, 3); });
            acc_last = accA;
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) zlag[o] = acc_last[0][L::HEAD_SLOT + o];
#pragma unroll
        for (int st = 0; st < 9; ++st) gate_stage(16 * NF32, st, acc_last, 3 * NF32);   // the remainder units (RJ <= 4)
    }

    // ---- the step in two segments, for the ping-pong kernels (split_kernels.h: prnn_flip_pp_kernel) -----------------
    // A SIMD issues one vector instruction per ~4 cycles and an MFMA holds the matrix pipe for 32: run back to back by
    // one wave, the 95 MFMAs and the ~470 VALU instructions of a step add up (measured: tools/microbench/issue_model,
    // profiles/r02_issue_model.txt).  Two waves of a SIMD that alternate - one in its MFMA segment while the other
    // is in its VALU segment, workgroup barrier between segments - overlap the two almost completely.
    // mfma_seg: accumulators <- bias / one-hot rows + the six products (no VALU work besides the B-register moves).
    static __device__ __forceinline__ void mfma_seg(const char* lds, int sig, const unsigned (&R)[3][NR], f32x16 (&acc)[NT], int lane) {
        static_assert(MODE != 0, "ping-pong kernels use the K-packed layouts (bias / input rows in the accumulator table)");
        const int hh = lane >> 5;
        asm volatile("" ::: "memory");
        const u32x4* av = reinterpret_cast<const u32x4*>(lds + L::OFF_A) + lane;
        constexpr int ORD[6][2] = {{2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};
        constexpr int NK = KS;
        const f32x16* ci = reinterpret_cast<const f32x16*>(lds + L::OFF_CI - L::LSHIFT) + (size_t)sig * NT * 2 + hh;
        auto a_index = [&](int t, int k) {
            if constexpr (MODE == 1) return (k * NT + t) * 64;
            else return k < 6 * NQ ? ((t * 3 + ORD[k / NQ][0]) * NQ + k % NQ) * 64 : (NT * 3 * NQ + t) * 64;
        };
        auto b_reg = [&](int k, int j) -> unsigned {
            if constexpr (MODE == 1) {
                const int f = 4 * k + j;
                return f < 6 * NRM ? R[ORD[f / NRM][1]][f % NRM] : 0u;
            } else {
                if constexpr (MODE == 2) {
                    if (k >= 6 * NQ) return j < 3 ? R[j][NRM] : 0u;
                }
                return R[ORD[k / NQ][1]][4 * (k % NQ) + j];
            }
        };
        // all NT tiles per k-step: one B quad per k-step, fragments of k-step k+1 in flight while k-step k multiplies
        u32x4 cur[NT], nxt[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = ci[t * 2];
#pragma unroll
        for (int t = 0; t < NT; ++t) cur[t] = av[a_index(t, 0)];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (k + 1 < NK) {
#pragma unroll
                for (int t = 0; t < NT; ++t) nxt[t] = av[a_index(t, k + 1)];
            }
            const u32x4 bq = {b_reg(k, 0), b_reg(k, 1), b_reg(k, 2), b_reg(k, 3)};
            const bf16x8 b = __builtin_bit_cast(bf16x8, bq);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[t]), b, acc[t], 0, 0, 0);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < NT; ++t) cur[t] = nxt[t];
        }
    }
    // gates_seg: new state of this lane's units from the accumulators (pure VALU + the candidate's input table)
    static __device__ __forceinline__ void gates_seg(const char* lds, int sig, const f32x16 (&acc)[NT], float (&h)[NU], int lane) {
        const int hh = lane >> 5;
        const float* xc = reinterpret_cast<const float*>(lds + L::OFF_XC - L::LSHIFT) + (size_t)((sig * 2 + hh) * L::NUP);
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            float ar, au, ac;
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
            const float rg = Act<float>::sigmoid_scaled(ar);
            const float ug = Act<float>::sigmoid_scaled(au);
            const float cc = Act<float>::tanh_scaled(xc[e] + rg * ac);
            h[e] = cc + ug * (h[e] - cc);
        }
    }

    // output head rows (row 0: softmax logit difference), reduced over the two lane halves
    static __device__ __forceinline__ void head(const char* lds, const float (&h)[NU], int lane, float (&z)[NOUT]) {
        const int hh = lane >> 5;
        asm volatile("" ::: "memory");
        const float* wd = reinterpret_cast<const float*>(lds + L::OFF_WD - L::LSHIFT) + hh * L::NUP * NOUT;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) z[o] = 0.0f;
#pragma unroll
        for (int e = 0; e < NU; ++e)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) z[o] = fmaf(h[e], wd[e * NOUT + o], z[o]);
        const float* bd = reinterpret_cast<const float*>(lds + L::OFF_BD - L::LSHIFT);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) z[o] += __shfl_xor(z[o], 32) + bd[o];
    }
};

}  // namespace rnnwf
