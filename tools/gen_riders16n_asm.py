#!/usr/bin/env python3
"""Generates rnnwavefunctions_amd/csrc/split_riders16n_asm.h: the whole wave-step of the bf16x3 flip pass at 37..52 units on
v_mfma_f32_16x16x32_bf16 (csrc/split16_core.h: S16nLayout) as ONE hand-scheduled inline-asm block.

The 69..100-unit form (tools/gen_riders16_asm.py) scaled down to 13 units per lane group: one wave per SIMD, two sets of 16 chains
per wave, every A fragment read once for both sets, VALU work list-scheduled into the slots behind the MFMAs.  What is new here:
  * 10 tiles x 10 k-steps x 2 sets = 200 MFMAs: k-steps of a tile = 6 products over the units j = 0..7 of every lane group (an
    "octet"), 3 k-steps that carry TWO products each over the units j = 8..11 (4 units x 2 products fill the 8 K entries of a lane
    group: (w1|w1) x (h2|h1), (w2|w1) x (h1|h3), (w2|w3) x (h2|h1)), and the special k-step of unit j = 12 (six products in
    six entries); the whole image (70 fragments of 1 KB + tables) is resident in LDS;
  * the accumulators live in VGPRs and the gates work IN PLACE on them (no v_accvgpr_read, no gate temporaries but the candidate
    input rows); AGPRs hold only the fragment ring;
  * tiles go in four groups - units j = 0..3, j = 4..7, j = 8..11, j = 12 - and the gates of a group ride on the MFMAs of the
    next one; the re-split of the state rides on the first group.

Register map (the kernel around the block keeps v0..v47; the state travels as two pinned f32x16 operands):
  v48..v63    temporaries: split / candidate input rows of two gate batches
  v64..v95    h: set X, unit 4 j + g at 64 + 13 X + j (in / out); v90 + X: head logit of the state that ENTERED the step (out)
  v96..v175   accumulators: tile t, set X, row r at 96 + 8 t + 4 X + r
  v176..v223  per set (24 registers): state parts of the octet [3 parts][4 pairs], QA = (h2 | h1) and QB = (h1 | h3) of the pairs
              (8,9), (10,11), RS = the special k-step's quad
  v224..v247  split residuals
  a0..        fragment ring (RING quads)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_riders_asm as G   # noqa: E402
from gen_riders_asm import Ins, valu, trans, V, VQ, AQ   # noqa: E402

NT, NJ, NFR, XCP = 10, 13, 7, 16
RING = int(os.environ.get("RIDERS_RING", "16"))
assert RING <= 64
# tile groups in issue order: the mixed tile (unit j = 12 + head row) with the tiles of j = 0..3, then j = 4..7, then j = 8..11; the
# gates of a group ride on the MFMAs of the next one, those of the last group (eight independent instances) are the step's tail
GROUPS = [[9, 0, 2, 4], [1, 3, 5], [6, 7, 8]]
# k-steps of a tile in issue order: (kind, fragment index, B quad): products that need h1 only first, then h2, then h3
KSEQ = [("O", 2, ("R", 0)), ("O", 1, ("R", 0)), ("O", 0, ("R", 0)), ("O", 1, ("R", 1)), ("O", 0, ("R", 1)), ("H", 3, ("QA",)), ("H", 5, ("QA",)),
        ("O", 0, ("R", 2)), ("H", 4, ("QB",)), ("P", 6, ("RS",))]
KS = len(KSEQ)


def layout():
    off_ci = NT * NFR * 1024
    off_xc = off_ci + 2 * NT * 4 * 16
    off_wd = off_xc + 2 * 4 * XCP * 4
    off_bd = off_wd + 4 * XCP * 4
    return dict(OFF_CI=off_ci, OFF_XC=off_xc, OFF_WD=off_wd, OFF_BD=off_bd, BYTES=off_bd + 16)


def vh(X, j): return 64 + NJ * X + j
def vacc(t, X, r): return 96 + 8 * t + 4 * X + r
def vR(X, p, i): return 176 + 24 * X + 4 * p + i
def vQA(X, i): return 176 + 24 * X + 12 + i                # h2 (8,9), h2 (10,11), h1 (8,9), h1 (10,11)
def vQB(X, i): return 176 + 24 * X + 16 + i                # h1 (8,9), h1 (10,11), h3 (8,9), h3 (10,11)
def vRS(X, i): return 176 + 24 * X + 20 + i
def aring(r): return 4 * r


def gate_slots(j):
    """(tile, row) of the r, u and candidate pre-activations of unit j of a lane group."""
    if j < 8:
        return [(2 * gate + j // 4, j % 4) for gate in range(3)]
    if j < 12:
        return [(6 + gate, j - 8) for gate in range(3)]
    return [(9, gate) for gate in range(3)]


class Step16n(G.Step):
    def __init__(self, budget=12):
        G.Step.__init__(self, 1, "tile", budget)
        self.L = layout()

    def frag_addr(self, t, f):
        off = (t * NFR + f) * 1024
        return "%%[l%d] offset:%d" % (off >> 16, off & 0xffff)

    def b_quad(self, bq, X):
        if bq[0] == "R":
            return vR(X, bq[1], 0)
        return {"QA": vQA, "QB": vQB, "RS": vRS}[bq[0]](X, 0)

    def frag_list(self):
        """(tile, k-step) in issue order, k-major within a group; each feeds the MFMAs of both chain sets."""
        return [(t, k) for grp in GROUPS for k in range(KS) for t in grp]

    # ---- riders -------------------------------------------------------------------------------------------------------
    def split_riders(self):
        """Everything the products need beyond the octet's h1 quads (those are converted in front of the first MFMA), in the order of
        first use: the octet's h2, the pairs' h1 and h2, the octet's h3, the pairs' h3, the special unit."""
        q = []
        T = list(range(48, 64))
        res0 = lambda X, i: 224 + 6 * X + i                   # pair i (0..3 octet, 4..5 the pairs (8,9), (10,11)) of set X
        res1 = lambda X, i: 236 + 6 * X + i
        OCT = [(X, i) for X in range(2) for i in range(4)]
        TL = [(X, i) for X in range(2) for i in (4, 5)]
        h1reg = lambda X, i: vR(X, 0, i) if i < 4 else vQA(X, 2 + i - 4)
        h2reg = lambda X, i: vR(X, 1, i) if i < 4 else vQA(X, i - 4)
        h3reg = lambda X, i: vR(X, 2, i) if i < 4 else vQB(X, 2 + i - 4)

        def level2(P):
            for n, (X, i) in enumerate(P): q.append(valu("v_lshlrev_b32", T[n], "16", h1reg(X, i)))
            for n, (X, i) in enumerate(P): q.append(valu("v_and_b32", T[8 + n], "0xffff0000", h1reg(X, i)))
            for n, (X, i) in enumerate(P): q.append(valu("v_sub_f32", res0(X, i), vh(X, 2 * i), T[n]))
            for n, (X, i) in enumerate(P): q.append(valu("v_sub_f32", res1(X, i), vh(X, 2 * i + 1), T[8 + n]))
            for n, (X, i) in enumerate(P): q.append(valu("v_cvt_pk_bf16_f32", h2reg(X, i), res0(X, i), res1(X, i)))

        def level3(P):
            for n, (X, i) in enumerate(P): q.append(valu("v_lshlrev_b32", T[n], "16", h2reg(X, i)))
            for n, (X, i) in enumerate(P): q.append(valu("v_and_b32", T[8 + n], "0xffff0000", h2reg(X, i)))
            for n, (X, i) in enumerate(P): q.append(valu("v_sub_f32", T[n], res0(X, i), T[n]))
            for n, (X, i) in enumerate(P): q.append(valu("v_sub_f32", T[8 + n], res1(X, i), T[8 + n]))
            for n, (X, i) in enumerate(P): q.append(valu("v_cvt_pk_bf16_f32", h3reg(X, i), T[n], T[8 + n]))

        pro = len(q)
        level2(OCT)
        self.n_prologue = len(q) - pro                         # issued in front of the first MFMA (the first fragments are in flight)
        for X, i in TL:                                        # the pairs' h1, once in QA and once in QB
            q.append(valu("v_cvt_pk_bf16_f32", vQA(X, 2 + i - 4), vh(X, 2 * i), vh(X, 2 * i + 1)))
        for X, i in TL:
            q.append(valu("v_cvt_pk_bf16_f32", vQB(X, i - 4), vh(X, 2 * i), vh(X, 2 * i + 1)))
        level2(TL)
        level3(OCT)
        level3(TL)
        # unit 12 of each set: parts in both halves of a register, then the special k-step's B quad
        Q = [[T[0], T[1], T[2]], [T[3], T[4], T[5]]]
        Xr = [vh(0, NJ - 1), vh(1, NJ - 1)]
        for X in range(2): q.append(valu("v_cvt_pk_bf16_f32", Q[X][0], Xr[X], Xr[X]))
        for X in range(2): q.append(valu("v_lshlrev_b32", T[6 + X], "16", Q[X][0]))
        for X in range(2): q.append(valu("v_sub_f32", T[6 + X], Xr[X], T[6 + X]))
        for X in range(2): q.append(valu("v_cvt_pk_bf16_f32", Q[X][1], T[6 + X], T[6 + X]))
        for X in range(2): q.append(valu("v_lshlrev_b32", T[8 + X], "16", Q[X][1]))
        for X in range(2): q.append(valu("v_sub_f32", T[8 + X], T[6 + X], T[8 + X]))
        for X in range(2): q.append(valu("v_cvt_pk_bf16_f32", Q[X][2], T[8 + X], T[8 + X]))
        # K entries 0..5 = {h1, h2, h3, h1, h2, h1} (A side {w1, w1, w1, w2, w2, w3}), two per register
        for X in range(2):
            q.append(valu("v_bfi_b32", vRS(X, 0), "s44", Q[X][0], Q[X][1]))
            q.append(valu("v_bfi_b32", vRS(X, 1), "s44", Q[X][2], Q[X][0]))
            q.append(valu("v_bfi_b32", vRS(X, 2), "s44", Q[X][1], Q[X][0]))
            q.append(valu("v_mov_b32", vRS(X, 3), "0"))
        return q

    def gate_batch(self, units, gx, tag, tile_done):
        """Gates of the (set, unit) instances in `units` (all of one tile group), stage by stage, in place on the accumulators.
        gx: registers for the candidate input rows, one per instance; instances come as runs of consecutive j per set."""
        n = len(units)
        ar = [vacc(gate_slots(j)[0][0], X, gate_slots(j)[0][1]) for X, j in units]
        au = [vacc(gate_slots(j)[1][0], X, gate_slots(j)[1][1]) for X, j in units]
        ac = [vacc(gate_slots(j)[2][0], X, gate_slots(j)[2][1]) for X, j in units]
        rdy = max(tile_done[t] for X, j in units for t, r in gate_slots(j)) + 3
        q = []
        # candidate input rows: one read per run of consecutive j of a set
        runs = []
        for n0, (X, j) in enumerate(units):
            if runs and runs[-1][0] == X and runs[-1][1] + runs[-1][2] == j:
                runs[-1][2] += 1
            else:
                runs.append([X, j, 1, n0])
        for X, j0, cnt, n0 in runs:
            assert cnt in (1, 4)
            op = "ds_read_b128 %s" % VQ(gx[n0]) if cnt == 4 else "ds_read_b32 %s" % V(gx[n0])
            i = Ins("%s, %%[xc%d] offset:%d" % (op, X, j0 * 4), 2, (), ["v%d" % (gx[n0] + x) for x in range(cnt)], "lds")
            i.lds_tag = ("xc", tag, X, j0)
            q.append(i)
        need = {}
        for X, j0, cnt, n0 in runs:
            for x in range(cnt):
                need[n0 + x] = ("xc", tag, X, j0)
        U = range(n)
        q += [trans("v_exp_f32", ar[u], ar[u]) for u in U]
        q += [trans("v_exp_f32", au[u], au[u]) for u in U]
        q += [valu("v_add_f32", ar[u], "1.0", ar[u]) for u in U]
        q += [valu("v_add_f32", au[u], "1.0", au[u]) for u in U]
        q += [trans("v_rcp_f32", ar[u], ar[u]) for u in U]
        q += [trans("v_rcp_f32", au[u], au[u]) for u in U]
        for u in U:
            i = valu("v_fma_f32", ac[u], ar[u], ac[u], gx[u])
            i.lds_need = need[u]
            q.append(i)
        q += [trans("v_exp_f32", ac[u], ac[u]) for u in U]
        q += [valu("v_add_f32", ac[u], "1.0", ac[u]) for u in U]
        q += [trans("v_rcp_f32", ac[u], ac[u]) for u in U]
        q += [valu("v_fma_f32", ac[u], "2.0", ac[u], "-1.0") for u in U]
        q += [valu("v_sub_f32", ar[u], vh(*units[u]), ac[u]) for u in U]
        q += [valu("v_fma_f32", vh(*units[u]), au[u], ar[u], ac[u]) for u in U]
        for ins in q:
            ins.ready = max(ins.ready, rdy)
        return q

    # ---- the schedule -------------------------------------------------------------------------------------------------
    def build(self):
        frags = self.frag_list()
        tile_done, first_of_tile = {}, {}
        for f, (t, k) in enumerate(frags):
            tile_done[t] = 2 * f + 1                               # index of the tile's last MFMA
            first_of_tile.setdefault(t, 2 * f)

        def lds_read(i):
            t, k = frags[i]
            ins = Ins("ds_read_b128 %s, %s" % (AQ(aring(i % RING)), self.frag_addr(t, KSEQ[k][1])), 2, (),
                      ["a%d" % (aring(i % RING) + n) for n in range(4)], "lds")
            ins.lds_tag = ("A", i)
            return ins

        def ci_read(t, X):
            ins = Ins("ds_read_b128 %s, %%[ci%d] offset:%d" % (VQ(vacc(t, X, 0)), X, t * 64), 2, (),
                      ["v%d" % vacc(t, X, r) for r in range(4)], "lds")
            ins.lds_tag = ("ci", t, X)
            return ins

        split_q = self.split_riders()
        last_g0 = max(tile_done[t] for t in GROUPS[0])
        T = list(range(48, 64))
        rem_q = self.gate_batch([(X, 12) for X in range(2)], [248, 249], "g3", tile_done)
        # head logit of the state that entered the step: row 3 of the mixed tile
        for X in range(2):
            mv = valu("v_mov_b32", 90 + X, vacc(9, X, 3))
            mv.ready = rem_q[-1].ready
            rem_q.insert(0, mv)
        batches = [self.gate_batch([(X, j) for X in range(2) for j in range(0, 4)], T[0:8], "g0", tile_done),
                   rem_q,
                   self.gate_batch([(X, j) for X in range(2) for j in range(4, 8)], T[8:16], "g1", tile_done),
                   self.gate_batch([(X, j) for X in range(2) for j in range(8, 12)], T[0:8], "g2", tile_done)]
        for q in batches:
            for ins in q:
                ins.ready = max(ins.ready, last_g0 + 1)               # the split's temporaries are the batches' rows: no gate inside group 0

        P = self.emit
        self.raw("s_mov_b32 s44, 0xffff", 1, "salu")
        self.raw("s_waitcnt lgkmcnt(0)", 1, "wait")
        for t in GROUPS[0]:
            for X in range(2):
                P(ci_read(t, X))
        n_l0 = min(RING - 1, len(frags))
        for i in range(n_l0):
            P(lds_read(i))
        next_l = n_l0
        written = set()
        for X in range(2):
            for i in range(4):
                ins = valu("v_cvt_pk_bf16_f32", vR(X, 0, i), vh(X, 2 * i), vh(X, 2 * i + 1))
                P(ins)
                written |= ins.writes
        if os.environ.get("RIDERS_PROLOGUE", "1") == "1":
            for ins in split_q[:self.n_prologue]:
                P(ins)
                written |= ins.writes
            del split_q[:self.n_prologue]
        ci_pending = [(t, X) for grp in GROUPS[1:] for t in grp for X in range(2)]
        streams = [split_q] + batches
        rows_of = [None, 0, 2, 1, 0]                           # which temporaries a batch keeps its candidate rows in
        heads = [0] * len(streams)

        def stream_done(si): return heads[si] >= len(streams[si])

        m = -1
        for f, (t, k) in enumerate(frags):
            for X in range(2):
                m += 1
                bq = self.b_quad(KSEQ[k][2], X)
                areg = aring(f % RING)
                d = vacc(t, X, 0)
                mf = Ins("v_mfma_f32_16x16x32_bf16 v[%d:%d], %s, %s, v[%d:%d]" % (d, d + 3, AQ(areg), VQ(bq), d, d + 3),
                         G.MFMA_ISSUE, ["v%d" % (bq + n) for n in range(4)], (), "mfma")
                if X == 0:
                    mf.lds_need = ("A", f)
                if m == first_of_tile[t] + X:
                    self.wait_lds(("ci", t, X))
                for n in range(4):
                    assert "v%d" % (bq + n) in written, ("B quad not written before MFMA", m, t, k, X, bq)
                P(mf)
                self.stats["mfma"] += 1
                used = 0
                if X == 0 and f >= 1 and next_l < len(frags) and next_l <= f - 1 + RING:
                    # refill the slot the PREVIOUS fragment's MFMAs consumed (they issued before this one)
                    ins = lds_read(next_l)
                    P(ins)
                    used += ins.cost
                    next_l += 1
                while ci_pending and first_of_tile[ci_pending[0][0]] <= m + 24:
                    ins = ci_read(*ci_pending.pop(0))
                    P(ins)
                    used += ins.cost
                # ---- riders: the split alone while it lasts, then the gate batches in order, at most two at a time (their rows
                # alternate between the two halves of the temporaries)
                if not stream_done(0):
                    active = [0]
                else:
                    active = []
                    for si in range(1, len(streams)):
                        if stream_done(si) or len(active) >= 3:
                            continue
                        if any(not stream_done(sj) and rows_of[sj] == rows_of[si] for sj in range(1, si)):
                            continue                           # an earlier, unfinished batch holds the same row registers
                        active.append(si)
                progress = True
                while used < self.budget and progress:
                    progress = False
                    for si in active:
                        if stream_done(si) or used >= self.budget:
                            continue
                        ins = streams[si][heads[si]]
                        if ins.ready > m:
                            continue
                        n0 = len(self.out)
                        P(ins)
                        used += sum(x.cost for x in self.out[n0:])
                        written |= ins.writes
                        heads[si] += 1
                        self.stats["riders"] += 1
                        progress = True
        assert stream_done(0), "the split must finish inside the first tile group"
        assert not ci_pending and next_l == len(frags)
        # tail: what is left of the batches (batch 2 shares its rows with batch 0: in order), then the unit-12 gates
        left = []
        for si in range(1, len(streams)):
            left += streams[si][heads[si]:]
        # the last group's gates read accumulators of tiles whose last MFMA has just issued: sixteen wait states first
        self.raw("s_nop 7", 32, "nop")
        self.raw("s_nop 7", 32, "nop")
        for ins in left:
            P(ins)
        self.raw("s_waitcnt lgkmcnt(0)", 1, "wait")
        self.stats["tail_left"] = len(left)
        return self.out


HEADER = '''// GENERATED by tools/gen_riders16n_asm.py - do not edit (edit the generator).
// The whole wave-step of the bf16x3 flip pass at 37..52 units on v_mfma_f32_16x16x32_bf16 (split16_core.h: S16nLayout) as one
// hand-scheduled asm block: 200 MFMAs of 16 cycles, two chain sets per wave, every A fragment read once for both, accumulators in
// VGPRs with the gates in place.
#pragma once
#include "split16_core.h"

namespace rnnwf {

template <int NOUT> struct Riders16nStepAsm { static constexpr bool kAvailable = false; };

'''


def struct_text(budget):
    st = Step16n(budget)
    st.build()
    L = st.L
    clob = ["v%d" % n for n in list(range(48, 64)) + list(range(96, 250))] + ["a%d" % n for n in range(4 * RING)] + ["s44", "memory"]
    clobs = ", ".join('"%s"' % c for c in clob)
    body = "\\n\\t".join(i.text for i in st.out if i.text)
    n_ins = sum(1 for i in st.out if i.text)
    issue = sum(i.cost for i in st.out)
    tmpl = '''// @NINS@ instructions, @NMFMA@ MFMAs; issue-cost model: @ISSUE@ cycles per wave-step (matrix pipe: @PIPE@); riders left for the tail: @LEFT@
template <> struct Riders16nStepAsm<1> {
    static constexpr bool kAvailable = true;
    using L = S16nLayout<1>;
    static_assert(L::OFF_CI == @CI@ && L::OFF_XC == @XC@ && L::NT == 10 && L::NFR == 7 && L::XCP == 16 && L::NJ == 13,
                  "generated for another layout: re-run tools/gen_riders16n_asm.py");
    // h0, h1: this lane's 2 x 13 state values (set X, unit 4 j + g at entry 13 X + j of the 32) in, the new state out; entries 26 + X out:
    // the head logit of the state that ENTERED the step, per set.  ci0 / ci1, xc0 / xc1: LDS byte addresses of the lane group's
    // accumulator-table row of tile 0 and of its candidate-input row for the input spin of set 0 / 1.  l0, l1: LDS byte address of
    // this lane's 16 bytes of fragment 0, + 0 / 64 KB.
    static __device__ __forceinline__ void run(f32x16& h0, f32x16& h1, unsigned ci0, unsigned ci1, unsigned xc0, unsigned xc1, unsigned l0,
                                               unsigned l1) {
        asm volatile("@BODY@"
                     : "+{v[64:79]}"(h0), "+{v[80:95]}"(h1)
                     : [ci0] "v"(ci0), [ci1] "v"(ci1), [xc0] "v"(xc0), [xc1] "v"(xc1), [l0] "v"(l0), [l1] "v"(l1)
                     : @CLOB@);
    }
};
'''
    rep = {"@NINS@": n_ins, "@NMFMA@": st.stats["mfma"], "@ISSUE@": issue, "@PIPE@": 16 * st.stats["mfma"], "@LEFT@": st.stats["tail_left"],
           "@CI@": L["OFF_CI"], "@XC@": L["OFF_XC"], "@BODY@": body.replace("%%", "%"), "@CLOB@": clobs}
    for k, v in rep.items():
        tmpl = tmpl.replace(k, str(v))
    return tmpl, st


def pipe_model(out):
    """Cycles if the matrix pipe (16 per MFMA) or the issue port (8 per MFMA + the cost of what sits behind it) bounds every MFMA gap."""
    total, gap, seen = 0, 0, False
    for i in out:
        if i.kind == "mfma":
            if seen:
                total += max(16, 8 + gap)
            else:
                total += gap
            seen, gap = True, 0
        else:
            gap += i.cost
    return total + 8 + gap


def main():
    budget = int(os.environ.get("RIDERS_BUDGET", "8"))
    txt, st = struct_text(budget)
    print("gap model: %d cycles per wave-step" % pipe_model(st.out))
    out = HEADER + txt + "\n}  // namespace rnnwf\n"
    print("budget=%d ring=%d:" % (budget, RING), st.stats, "instructions", sum(1 for i in st.out if i.text), "issue cycles", sum(i.cost for i in st.out))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "rnnwavefunctions_amd", "csrc", "split_riders16n_asm.h")
    with open(path, "w") as f:
        f.write(out)
    print("wrote", path)


if __name__ == "__main__":
    main()
