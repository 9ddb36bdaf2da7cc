// split16_core.h - the bf16x3 flip pass at 69..100 units on v_mfma_f32_16x16x32_bf16 (round 3).
//
// Why a second MFMA shape: the riders pass is POWER-limited (the hand-scheduled 32x32x16 step spends 14.5 % fewer
// cycles than hipcc's and gets 3.5 % less time - the chip lowers its clock as the MFMAs pack closer).  On random data the chip
// holds a much higher clock under the 16x16x32 shape: the same flops and the same ~1 000 VALU riders per wave-step run 14.5 %
// MORE cycles and 6 - 11 % LESS wall time (profiles/r03_issue_model_rid.txt: 2.2 GHz against 1.76 GHz).
//
// Shapes.  A wave still owns 32 chains, as two SETS of 16 (set X: chains 16 X + (lane & 15)); lane (c = lane & 15, g = lane >> 4).
// Lane group g owns, for both of its chains, the units u = 4 j + g, j = 0..24 - the ownership of the f32 engine (layout.h), so the
// base pass's checkpoints [KT16][64] are read as they lie.  C/D row 4 g + r of a 16-row tile therefore carries a unit of group g:
//     tile 6 b + 2 gate + tau (block b = 0..2, gate r / u / c, tau = 0, 1):  row 4 g + r  <->  unit 4 (8 b + 4 tau + r) + g
//     tile 18 ("mixed"):  row 4 g + r, r < 3: gate r of unit 96 + g;  r = 3: the head row (logit difference), in every group
// and the new state is exactly the next step's B operand: group g supplies the K entries 8 g .. 8 g + 7 of a k-step, which are its
// own units j = 8 o .. 8 o + 7 of octet o.  K-steps of a tile: 6 products x 3 octets (24 aligned units) + 1 special k-step whose
// entries 0..5 hold the six products {w1 h1, w1 h2, w1 h3, w2 h1, w2 h2, w3 h1} of the 25th unit: 19 k-steps x 19 tiles x 2 sets =
// 722 MFMAs of 16 cycles per wave-step (the 32x32x16 form: 380 of 32); every A fragment (1 KB) feeds both sets.
// Image: regular fragments [tile][part][octet], special fragments [tile], then the tables; parts 0, 1 and the special fragments
// live in LDS (133 KB), the w3 fragments (57 KB) are read through L2 as before.
#pragma once
#include "split_core.h"

namespace rnnwf {

template <int NOUT_ = 1>
struct S16Layout {
    static constexpr int NOUT = NOUT_;
    static constexpr int NJ = 25, NJA = 24, NOCT = 3;       // units per lane group and chain; aligned ones; octets per part
    static constexpr int NT = 19, NTF = 18;                 // tiles; full tiles
    static constexpr int HP = 100;
    static constexpr int KS = 6 * NOCT + 1;                 // k-steps per tile
    static constexpr int XCP = 28;                          // padded row of the candidate-input table (16-byte rows)
    static constexpr size_t OFF_A = 0;                                              // [NT][3][NOCT][64] x 16 B
    static constexpr size_t OFF_ASP = (size_t)NT * 3 * NOCT * 1024;                 // [NT][64] x 16 B
    static constexpr size_t OFF_CI = OFF_ASP + (size_t)NT * 1024;                   // [2 sigma][NT][4 g][4 r] f32
    static constexpr size_t OFF_XC = OFF_CI + (size_t)2 * NT * 4 * 16;              // [2 sigma][4 g][XCP] f32 (scaled)
    static constexpr size_t OFF_WD = OFF_XC + (size_t)2 * 4 * XCP * 4;              // [4 g][XCP] f32 head weights (logit difference)
    static constexpr size_t OFF_BD = OFF_WD + (size_t)4 * XCP * 4;                  // [4] f32
    static constexpr size_t BYTES = OFF_BD + 16;
    // LDS: fragments of parts 0, 1 compacted to [NT][2][NOCT], then everything from OFF_ASP on
    static constexpr size_t LDS_REG = (size_t)NT * 2 * NOCT * 1024;
    static constexpr size_t LSHIFT = OFF_ASP - LDS_REG;
    static constexpr size_t LDS_BYTES = BYTES - LSHIFT;
    static_assert(LDS_BYTES <= 160 * 1024, "the resident part of the image must fit LDS");
    // tile -> (gate, j of row r) helpers
    static constexpr int tile_of(int b, int gate, int tau) { return 6 * b + 2 * gate + tau; }
};

// The same form at 37..52 units (13 units per lane group; tools/gen_riders16n_asm.py, split_riders16n_asm.h).  Tiles: t = 2 gate + tau
// (tau = 0, 1): row 4 g + r <-> gate of unit 4 (4 tau + r) + g (j = 0..7); t = 6 + gate: unit 4 (8 + r) + g (j = 8..11); tile 9: row
// 4 g + r, r < 3: gate r of unit 48 + g (j = 12), r = 3: the head row.  K-steps of a tile (10): the six products over the "octet"
// j = 0..7 (fragments 0..2 = weight parts w1..w3 of the octet), three k-steps that carry TWO products each over j = 8..11 (K entries
// 0..3 | 4..7 of a lane group: fragment 3 = (w1 | w1) against the state quad (h2 | h1), fragment 4 = (w2 | w1) against (h1 | h3),
// fragment 5 = (w2 | w3) against (h2 | h1)), and the special k-step of j = 12 (fragment 6: {w1, w1, w1, w2, w2, w3} against
// {h1, h2, h3, h1, h2, h1}).  10 tiles x 10 k-steps x 2 sets = 200 MFMAs per wave-step; 70 fragments of 1 KB + tables, all in LDS.
template <int NOUT_ = 1>
struct S16nLayout {
    static constexpr int NOUT = NOUT_;
    static constexpr int NJ = 13, NT = 10, NFR = 7, KS = 10, HP = 52;
    static constexpr int XCP = 16;                          // padded row of the candidate-input table (64-byte rows)
    static constexpr size_t OFF_A = 0;                                              // [NT][NFR][64] x 16 B
    static constexpr size_t OFF_CI = (size_t)NT * NFR * 1024;                       // [2 sigma][NT][4 g][4 r] f32
    static constexpr size_t OFF_XC = OFF_CI + (size_t)2 * NT * 4 * 16;              // [2 sigma][4 g][XCP] f32 (scaled)
    static constexpr size_t OFF_WD = OFF_XC + (size_t)2 * 4 * XCP * 4;              // [4 g][XCP] f32 head weights (logit difference)
    static constexpr size_t OFF_BD = OFF_WD + (size_t)4 * XCP * 4;                  // [4] f32
    static constexpr size_t BYTES = OFF_BD + 16;
    static_assert(BYTES <= 160 * 1024, "the image must fit LDS");
};

}  // namespace rnnwf
