// layout.h - shared (host + device) description of the weight image the kernels stage in LDS.
//
// Orientation (see DESIGN.md "Recurrent operand layout"): every recurrent step computes
//     D^T[row, chain] = sum_k Wt[row, k] * h[k, chain]
// with the 16x16x4 MFMA: A = Wt (rows = gate pre-activations), B = h^T (columns = 16 chains of one
// wave).  The C/D fragment of lane (c = lane & 15, q = lane >> 4), register r, is then exactly the
// B fragment the NEXT step needs (unit 16 m + 4 r + q <-> k-step 4 m + r, lane quarter q), so the
// hidden state never leaves registers and never crosses lanes.
//
// GRU tiles (NT = 3 NFULL + 1, hidden size padded to HP = 16 NFULL + 4):
//   tile g*NFULL + m (g = 0 r-gate, 1 u-gate, 2 candidate-hidden), m < NFULL : units 16m .. 16m+15
//   tile 3*NFULL ("mixed")  : register r = gate r of unit 16 NFULL + q  (r = 3 unused)
// K-steps: KT = 4 NFULL + 1 (k = 4 kt + q).
#pragma once
// Timing-only ablation paths (skip MFMAs / gates / head / stores: WRONG numbers) exist only in -DRNNWF_DIAGNOSTICS
// builds made by tools/; in the release library the conditions are compile-time false.
#ifdef RNNWF_DIAGNOSTICS
#define RNNWF_ABLATED(mask, bit) (((mask) & (bit)) != 0)
#else
#define RNNWF_ABLATED(mask, bit) false
#endif
#include <cstddef>
#include <cstdint>

namespace rnnwf {

// exp / log tables of the float64 models (device.h: exp_tab, log1p_tab; filled by pack.h: fill_f64_tables)
struct F64Tables {                       // layout of the table block in LDS (doubles)
    static constexpr int EXP2 = 0;       // [64]  2^(j/64)
    static constexpr int RCPC = 64;      // [64]  1 / c_j,  c_j = 1 + (j + 1/2) / 64
    static constexpr int LOGC = 128;     // [64]  log(c_j)
    static constexpr int COUNT = 192;
    static constexpr size_t BYTES = COUNT * 8;
};


constexpr int kChains = 16;  // chains (spin configurations) per wave: the N dimension of the 16x16x4 MFMA

template <typename T, int NFULL_, int NOUT_>
struct GruLayout {
    static constexpr int NFULL = NFULL_;
    static constexpr int NOUT = NOUT_;              // head rows: 1 (pRNN logit difference), 3 (cRNN: + 2 phase logits)
    static constexpr int HP = 16 * NFULL + 4;       // padded hidden size
    static constexpr int KT = 4 * NFULL + 1;        // k-steps of 4
    static constexpr int NT = 3 * NFULL + 1;        // 16-row output tiles
    static constexpr int VW = 16 / (int)sizeof(T);  // A values per 16-byte LDS vector
    static constexpr int NG = (KT - 1) / VW;        // full vectors per (tile, lane)
    // byte offsets of the sections, every one 16-byte aligned
    static constexpr size_t OFF_AVEC = 0;                                            // [NT][NG][64] x 16 B
    static constexpr size_t OFF_AREM = OFF_AVEC + (size_t)NT * NG * 64 * 16;         // [NT][64] T   (kt = KT-1)
    static constexpr size_t OFF_BINIT = OFF_AREM + (size_t)NT * 64 * sizeof(T);      // [3][NT][4 q][4 r] T
    static constexpr size_t SZ_BINIT_VARIANT = ((size_t)NT * 16 + 4) * sizeof(T);    // +4 T pad: de-alias banks
    static constexpr size_t OFF_XC = OFF_BINIT + 3 * SZ_BINIT_VARIANT;               // [3][NFULL+1][4][4] T
    static constexpr size_t SZ_XC_VARIANT = ((size_t)(NFULL + 1) * 16 + 4) * sizeof(T);
    static constexpr int WD_Q = ((NOUT * KT + 3) / 4) * 4;                           // head weights per lane quarter
    static constexpr size_t OFF_WD = OFF_XC + 3 * SZ_XC_VARIANT;                     // [4 q][KT][NOUT] T, WD_Q per q
    static constexpr size_t OFF_BD = OFF_WD + (size_t)4 * WD_Q * sizeof(T);          // [NOUT] T (padded to 32 B)
    static constexpr size_t BYTES = ((OFF_BD + 32 + 15) / 16) * 16;
    // above 100 units (float) / 68 (double) the image no longer fits the 160 KB of LDS: it stays in global memory, its A fragments
    // are read through L2 with buffer loads (GruCore::mfma_streamed), the small tables with plain loads
    static constexpr bool SPILL = BYTES > 160 * 1024;
    static constexpr size_t LDS_BYTES = SPILL ? 0 : BYTES;
};

// bf16x3 operand image of the COOPERATIVE base pass (gru_kernels.h: coop_base_pass_bf; f32 models, NFULL <= 3): the same tiles,
// rows and unit assignment as GruLayout, but the matrix products run on v_mfma_f32_16x16x32_bf16 with both operands held
// exactly as three bf16 parts (split_core.h explains the arithmetic): A fragments [NT][3 parts][NKS][64 lanes] x 16 B, lane
// (row = lane & 15, g = lane >> 4) holding the row's weights for the K entries 32 t + 8 g .. + 7 of k-step t.  K axis of one
// part: the units in groups of 16, group m = wave m's units, entry 4 q + r of the group = unit 16 m + 4 r + q - the four units a
// lane (c, q) of that wave produces are then four consecutive entries (one 8-byte store per part into the block's operand
// buffer); the remainder group holds its units 16 NFULL + q at entries 4 q.  The bias / input / head tables stay GruLayout's.
template <int NFULL_>
struct BaseBfLayout {
    static constexpr int NFULL = NFULL_;
    static constexpr int NW = NFULL + 1;                    // waves per block of 16 chains = unit groups
    static constexpr int KU = ((16 * NW + 31) / 32) * 32;   // K entries per part (padded to whole k-steps)
    static constexpr int NKS = KU / 32;                     // k-steps per part
    static constexpr int NO = KU / 8;                       // 16-byte octets per part and chain in the operand buffer
    static constexpr int NT = 3 * NFULL + 1;
    static constexpr size_t BYTES = (size_t)NT * 3 * NKS * 64 * 16;
    static constexpr size_t PB_BYTES = (size_t)3 * NO * 16 * 16;   // operand buffer of one block: [part][octet][chain] x 16 B
    static constexpr int NB = 3;                            // blocks of 16 chains per workgroup (share one image in LDS)
};

// Stacked GRU layers above the first (tf.nn.rnn_cell.MultiRNNCell, 1DTFIM/RNNwavefunction.py:32):
// the input x is the new state of the layer below - already a B fragment - so one step is two blocks of
// products into one set of accumulators,
//   X block (K over x): rows r, u, y = x Wci + bci          H block (K over h): rows r, u, q = h Wch + bch
// each shaped like the first layer's image (NT = 3 NFULL + 1 tiles: two/three full groups + the mixed tile, whose
// register slots are 0: r, 1: u, 2: q (H block only), 3: y (X block only)).  Accumulator tiles: r | u | q | y | mixed.
template <int NFULL_, typename T = float>
struct UpperLayout {
    static constexpr int NFULL = NFULL_;
    static constexpr int KT = 4 * NFULL + 1;
    static constexpr int NT = 3 * NFULL + 1;        // tiles per block
    static constexpr int NT2 = 4 * NFULL + 1;       // accumulator tiles
    static constexpr int VW = 16 / (int)sizeof(T);  // A values per 16-byte LDS vector
    static constexpr int NG = (KT - 1) / VW;        // full vectors per (tile, lane)
    static constexpr size_t SZ_AVEC = (size_t)NT * NG * 64 * 16;
    static constexpr size_t SZ_AREM = (size_t)NT * 64 * sizeof(T);
    static constexpr size_t OFF_AX = 0;                         // [NT][NG][64] x 16 B
    static constexpr size_t OFF_AXR = OFF_AX + SZ_AVEC;         // [NT][64] T (kt = KT-1)
    static constexpr size_t OFF_AH = OFF_AXR + SZ_AREM;
    static constexpr size_t OFF_AHR = OFF_AH + SZ_AVEC;
    static constexpr size_t OFF_B = OFF_AHR + SZ_AREM;          // [NT2][4 q][4 r] T
    static constexpr size_t BYTES = ((OFF_B + (size_t)NT2 * 16 * sizeof(T) + 15) / 16) * 16;
};

// Where the images of all layers exceed the 160 KB of LDS (three layers of 50 units: 171 KB; run_1dTFIM.py's width with
// num_layers = 3) the top layer's image stays in global memory (MlSpill<...>::value = 1): its A fragments are then read
// through L2 every step (67 KB per wave-step, every wave the same lines) - slower, but the configuration runs.
// Wider stacks (round 3: 53..100 units, float) spill as many of the top layers as it takes - all NL - 1 of them at 100 units,
// where one upper image is 250 KB; the first layer's image always fits.
template <int NFULL, int NL, typename T, int NOUT = 1>
struct MlSpill {
    static constexpr size_t L0 = GruLayout<T, NFULL, NOUT>::BYTES, UP = UpperLayout<NFULL, T>::BYTES;
    static constexpr int pick() {
        for (int s = 0; s < NL; ++s)
            if (L0 + (size_t)(NL - 1 - s) * UP <= 160 * 1024) return s;
        return NL - 1;
    }
    static constexpr int value = pick();
    static_assert(L0 <= 160 * 1024, "the first layer's image must fit LDS");
};

// row index inside a 16-row tile  <->  (lane quarter q of the C/D fragment, register r)
//   f32 16x16x4 : row = 4 q + r          f64 16x16x4 : row = q + 4 r
template <typename T> inline void row_to_qr(int row, int& q, int& r);
template <> inline void row_to_qr<float>(int row, int& q, int& r) { q = row >> 2; r = row & 3; }
template <> inline void row_to_qr<double>(int row, int& q, int& r) { q = row & 3; r = row >> 2; }

}  // namespace rnnwf
