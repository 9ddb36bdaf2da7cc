#!/usr/bin/env python3
"""Generates rnnwavefunctions_amd/csrc/split_riders16_asm.h: the whole wave-step of the bf16x3 flip pass at 69..100 units on
v_mfma_f32_16x16x32_bf16 (csrc/split16_core.h explains the layout) as ONE hand-scheduled inline-asm block.

Same machinery as tools/gen_riders_asm.py (the 32x32x16 form: emission with hazard padding, counted waits, fragment rings, rider
streams list-scheduled into the slots behind the MFMAs); what differs is the shape: 19 tiles of 16 rows x 19 k-steps x 2 chain sets =
722 MFMAs of 16 cycles, every A fragment read once for both sets, the state of set X in v(64 + 25 X) .. and the operand quads per set.

Register map:
  v48..v63    temporaries (split) / gate batch B (gc, gx)          v64..v113   h[2 sets][25] (in / out), v114 + X head logit (out)
  v116..v127  remainder units' gate temporaries                    v128..v199  R[2 sets][3 parts][12] state parts (MFMA B quads)
  v200..v207  RS[2 sets][4] special k-step's B quads               v208..v255  split residuals (pass 1) / gate batches A, B
  a0..a151    accumulators (tile t, set X) at 8 t + 4 X            a152..      LDS fragment ring (RING_L quads), then the
  streamed fragment ring (RING_S quads), up to a255
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_riders_asm as G   # noqa: E402
from gen_riders_asm import Ins, valu, trans, V, A, VQ, AQ   # noqa: E402

NT, NOCT, NJ, NJA = 19, 3, 25, 24
KS = 6 * NOCT + 1
LDS_REG = NT * 2 * NOCT * 1024
XCP = 28
ORD5 = G.ORD5
RING_L = int(os.environ.get("RIDERS_RING_L", "10"))       # quads; the two rings share a152..a255 (26 quads)
RING_S = int(os.environ.get("RIDERS_RING_S", "14"))
assert RING_L + RING_S <= 26


def layout():
    off_asp = NT * 3 * NOCT * 1024
    off_ci = off_asp + NT * 1024
    off_xc = off_ci + 2 * NT * 4 * 16
    off_wd = off_xc + 2 * 4 * XCP * 4
    off_bd = off_wd + 4 * XCP * 4
    total = off_bd + 16
    lshift = off_asp - LDS_REG
    return dict(OFF_ASP=off_asp, OFF_CI=off_ci, OFF_XC=off_xc, OFF_WD=off_wd, BYTES=total, LSHIFT=lshift)


def vh(X, j): return 64 + 25 * X + j
def vR(X, p, i): return 128 + 36 * X + 12 * p + i
def vRS(X, k): return 200 + 4 * X + k
def aacc(t, X, r): return 8 * t + 4 * X + r
def aringL(r): return 152 + 4 * r
def aringS(r): return 152 + 4 * RING_L + 4 * r


class Step16(G.Step):
    def __init__(self, order23="tile", budget=10):
        G.Step.__init__(self, 1, order23, budget)
        self.L = layout()

    def lds_frag_addr(self, t, kind, idx):
        if kind == "L":
            wp = ORD5[idx // NOCT][0]
            off = ((t * 2 + wp) * NOCT + idx % NOCT) * 1024
        else:
            off = LDS_REG + t * 1024
        return "%%[l%d] offset:%d" % (off >> 16, off & 0xffff)

    def klists(self):
        std = []
        for j in range(NOCT):
            std.append(("S", j))
            std += [("L", 5 * j + i) for i in range(5)]
        std.append(("P", 0))
        p1 = [("L", k) for k in range(2 * NOCT)]
        for j in range(NOCT):
            p1.append(("S", j))
            p1 += [("L", 2 * NOCT + 3 * j + i) for i in range(3)]
        p1.append(("P", 0))
        assert len(std) == KS and len(p1) == KS and sorted(x for x in p1 if x[0] == "L") == sorted(x for x in std if x[0] == "L")
        return p1, std

    def frag_list(self):
        """(tile, k-entry) in issue order; each feeds the MFMAs of both chain sets."""
        p1, std = self.klists()
        seq = [(t, k) for k in p1 for t in range(6)]                       # block 0: k-major (h2 / h3 appear while it runs)
        for b in (1, 2):
            tiles = range(6 * b, 6 * b + 6)
            if self.order23 == "tile":
                seq += [(t, k) for t in tiles for k in std]
            else:
                seq += [(t, k) for k in std for t in tiles]
        seq += [(18, k) for k in std]
        return seq

    def b_quad(self, k, X):
        kind, idx = k
        if kind == "L":
            return vR(X, ORD5[idx // NOCT][1], 4 * (idx % NOCT))
        if kind == "S":
            return vR(X, 0, 4 * idx)
        return vRS(X, 0)

    # ---- riders -------------------------------------------------------------------------------------------------------
    def split_riders(self):
        q = []
        T = list(range(48, 64))
        NP = NJA // 2                                           # pairs per set
        pairs = [(X, i) for X in range(2) for i in range(NP)]
        res0 = lambda X, i: 208 + 12 * X + i
        res1 = lambda X, i: 232 + 12 * X + i
        # h1 of everything but octet 0 (converted in front of the first MFMA)
        for o in range(1, NOCT):
            for X in range(2):
                for i in range(4 * o, 4 * o + 4):
                    q.append(valu("v_cvt_pk_bf16_f32", vR(X, 0, i), vh(X, 2 * i), vh(X, 2 * i + 1)))
        for o in range(NOCT):                                   # h2 of octet o, both sets: stage by stage over eight pairs
            P = [(X, i) for X in range(2) for i in range(4 * o, 4 * o + 4)]
            for n, (X, i) in enumerate(P): q.append(valu("v_lshlrev_b32", T[n], "16", vR(X, 0, i)))
            for n, (X, i) in enumerate(P): q.append(valu("v_and_b32", T[8 + n], "0xffff0000", vR(X, 0, i)))
            for n, (X, i) in enumerate(P): q.append(valu("v_sub_f32", res0(X, i), vh(X, 2 * i), T[n]))
            for n, (X, i) in enumerate(P): q.append(valu("v_sub_f32", res1(X, i), vh(X, 2 * i + 1), T[8 + n]))
            for n, (X, i) in enumerate(P): q.append(valu("v_cvt_pk_bf16_f32", vR(X, 1, i), res0(X, i), res1(X, i)))
        for o in range(NOCT):                                   # h3 of octet o
            P = [(X, i) for X in range(2) for i in range(4 * o, 4 * o + 4)]
            for n, (X, i) in enumerate(P): q.append(valu("v_lshlrev_b32", T[n], "16", vR(X, 1, i)))
            for n, (X, i) in enumerate(P): q.append(valu("v_and_b32", T[8 + n], "0xffff0000", vR(X, 1, i)))
            for n, (X, i) in enumerate(P): q.append(valu("v_sub_f32", T[n], res0(X, i), T[n]))
            for n, (X, i) in enumerate(P): q.append(valu("v_sub_f32", T[8 + n], res1(X, i), T[8 + n]))
            for n, (X, i) in enumerate(P): q.append(valu("v_cvt_pk_bf16_f32", vR(X, 2, i), T[n], T[8 + n]))
        # the 25th unit of each set: parts in both halves of a register, then the special k-step's B quad
        Q = [[T[0], T[1], T[2]], [T[3], T[4], T[5]]]
        Xr = [vh(0, NJA), vh(1, NJA)]
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

    def gate_batch16(self, b, X, regs, tile_done):
        gr, gu, gc, gx = regs
        U = range(8)
        tile = lambda gate, u: 6 * b + 2 * gate + u // 4

        def part(gate, g):
            rdy = max(tile_done[6 * b + 2 * gate], tile_done[6 * b + 2 * gate + 1]) + 3
            q = []
            for u in U:
                src = aacc(tile(gate, u), X, u % 4)
                i = Ins("v_accvgpr_read_b32 %s, %s" % (V(g[u]), A(src)), 4, ["a%d" % src], ["v%d" % g[u]], "valu", rdy)
                i.accread = True
                q.append(i)
            return q
        qr = part(0, gr)
        qr += [trans("v_exp_f32", gr[u], gr[u]) for u in U]
        qr += [valu("v_add_f32", gr[u], "1.0", gr[u]) for u in U]
        qr += [trans("v_rcp_f32", gr[u], gr[u]) for u in U]
        qu = part(1, gu)
        qu += [trans("v_exp_f32", gu[u], gu[u]) for u in U]
        qu += [valu("v_add_f32", gu[u], "1.0", gu[u]) for u in U]
        qu += [trans("v_rcp_f32", gu[u], gu[u]) for u in U]
        qc = part(2, gc)
        for w in range(2):
            i = Ins("ds_read_b128 %s, %%[xc%d] offset:%d" % (VQ(gx[4 * w]), X, (8 * b + 4 * w) * 4), 2, (), ["v%d" % (gx[4 * w] + n) for n in range(4)], "lds")
            i.lds_tag = ("xc", b, X, w)
            qc.append(i)
        for u in U:
            i = valu("v_fma_f32", gc[u], gr[u], gc[u], gx[u])
            i.lds_need = ("xc", b, X, u // 4)
            qc.append(i)
        qc += [trans("v_exp_f32", gc[u], gc[u]) for u in U]
        qc += [valu("v_add_f32", gc[u], "1.0", gc[u]) for u in U]
        qc += [trans("v_rcp_f32", gc[u], gc[u]) for u in U]
        qc += [valu("v_fma_f32", gc[u], "2.0", gc[u], "-1.0") for u in U]
        qc += [valu("v_sub_f32", gr[u], vh(X, 8 * b + u), gc[u]) for u in U]
        qc += [valu("v_fma_f32", vh(X, 8 * b + u), gu[u], gr[u], gc[u]) for u in U]
        return qr + qu + qc

    def remainder_gates16(self):
        gr, gu, gc, gx = [118, 119], [120, 121], [122, 123], [116, 117]
        q = []
        for X in range(2):
            i = Ins("ds_read_b32 %s, %%[xc%d] offset:%d" % (V(gx[X]), X, NJA * 4), 2, (), ["v%d" % gx[X]], "lds")
            i.lds_tag = ("xc", "rem", X)
            q.append(i)
        for g, r in ((gr, 0), (gu, 1), (gc, 2)):
            for X in range(2):
                rd = Ins("v_accvgpr_read_b32 %s, %s" % (V(g[X]), A(aacc(18, X, r))), 4, ["a%d" % aacc(18, X, r)], ["v%d" % g[X]])
                rd.accread = True
                q.append(rd)
        for X in range(2):
            rd = Ins("v_accvgpr_read_b32 %s, %s" % (V(114 + X), A(aacc(18, X, 3))), 4, ["a%d" % aacc(18, X, 3)], ["v%d" % (114 + X)])
            rd.accread = True
            q.append(rd)
        J = range(2)
        q += [trans("v_exp_f32", g[j], g[j]) for g in (gr, gu) for j in J]
        q += [valu("v_add_f32", g[j], "1.0", g[j]) for g in (gr, gu) for j in J]
        q += [trans("v_rcp_f32", g[j], g[j]) for g in (gr, gu) for j in J]
        for j in J:
            f = valu("v_fma_f32", gc[j], gr[j], gc[j], gx[j])
            f.lds_need = ("xc", "rem", j)
            q.append(f)
        q += [trans("v_exp_f32", gc[j], gc[j]) for j in J]
        q += [valu("v_add_f32", gc[j], "1.0", gc[j]) for j in J]
        q += [trans("v_rcp_f32", gc[j], gc[j]) for j in J]
        q += [valu("v_fma_f32", gc[j], "2.0", gc[j], "-1.0") for j in J]
        q += [valu("v_sub_f32", gr[j], vh(j, NJA), gc[j]) for j in J]
        q += [valu("v_fma_f32", vh(j, NJA), gu[j], gr[j], gc[j]) for j in J]
        return q

    # ---- the schedule -------------------------------------------------------------------------------------------------
    def build(self):
        frags = self.frag_list()
        nf = len(frags)
        tile_done, first_of_tile = {}, {}
        for f, (t, k) in enumerate(frags):
            tile_done[t] = 2 * f + 1                               # index of the tile's last MFMA
            first_of_tile.setdefault(t, 2 * f)
        lds_ops = [f for f, (t, k) in enumerate(frags) if k[0] != "S"]
        s_ops = [f for f, (t, k) in enumerate(frags) if k[0] == "S"]
        lpos = {f: i for i, f in enumerate(lds_ops)}
        spos = {f: i for i, f in enumerate(s_ops)}

        def lds_read(i):
            t, k = frags[lds_ops[i]]
            ins = Ins("ds_read_b128 %s, %s" % (AQ(aringL(i % RING_L)), self.lds_frag_addr(t, k[0], k[1])), 2, (),
                      ["a%d" % (aringL(i % RING_L) + n) for n in range(4)], "lds")
            ins.lds_tag = ("A", i)
            return ins

        def s_req(i):
            t, k = frags[s_ops[i]]
            soff = ((t * 3 + 2) * NOCT + k[1]) * 1024
            sreg = "s%d" % (40 + i % 4)
            mov = Ins("s_mov_b32 %s, 0x%x" % (sreg, soff), 1, kind="salu")
            ld = Ins("buffer_load_dwordx4 %s, %%[vo], %%[rs], %s offen" % (AQ(aringS(i % RING_S)), sreg), 6, (),
                     ["a%d" % (aringS(i % RING_S) + n) for n in range(4)], "vmem")
            ld.vm_tag = ("S", i)
            return [mov, ld]

        def ci_read(t, X):
            ins = Ins("ds_read_b128 %s, %%[ci%d] offset:%d" % (AQ(aacc(t, X, 0)), X, t * 64), 2, (),
                      ["a%d" % aacc(t, X, r) for r in range(4)], "lds")
            ins.lds_tag = ("ci", t, X)
            return ins

        regsA = (list(range(208, 216)), list(range(216, 224)), list(range(224, 232)), list(range(232, 240)))
        regsB = (list(range(240, 248)), list(range(248, 256)), list(range(48, 56)), list(range(56, 64)))
        split_q = self.split_riders()
        last_p1 = max(tile_done[t] for t in range(6))
        batches = []
        for b in range(3):
            for X in range(2):
                regs = regsA if (2 * b + X) % 2 == 0 else regsB
                q = self.gate_batch16(b, X, regs, tile_done)
                for ins in q:
                    ins.ready = max(ins.ready, last_p1 + 1)
                batches.append(q)
        rem_q = self.remainder_gates16()

        P = self.emit
        self.raw("s_mov_b32 s44, 0xffff", 1, "salu")
        self.raw("s_waitcnt lgkmcnt(0)", 1, "wait")
        for t in range(6):
            for X in range(2):
                P(ci_read(t, X))
        n_l0 = min(RING_L - 1, len(lds_ops))
        for i in range(n_l0):
            P(lds_read(i))
        next_l = n_l0
        written = set()
        for X in range(2):
            for i in range(4):
                ins = valu("v_cvt_pk_bf16_f32", vR(X, 0, i), vh(X, 2 * i), vh(X, 2 * i + 1))
                P(ins)
                written |= ins.writes
        next_s = 0
        ci_pending = [(t, X) for t in range(6, NT) for X in range(2)]
        streams = [split_q] + batches
        heads = [0] * len(streams)
        active = [0]

        def stream_done(si): return heads[si] >= len(streams[si])

        m = -1
        for f, (t, k) in enumerate(frags):
            for X in range(2):
                m += 1
                bq = self.b_quad(k, X)
                areg = aringS(spos[f] % RING_S) if k[0] == "S" else aringL(lpos[f] % RING_L)
                d = aacc(t, X, 0)
                mf = Ins("v_mfma_f32_16x16x32_bf16 a[%d:%d], %s, %s, a[%d:%d]" % (d, d + 3, AQ(areg), VQ(bq), d, d + 3),
                         G.MFMA_ISSUE, ["v%d" % (bq + n) for n in range(4)], (), "mfma")
                if X == 0:
                    if k[0] == "S":
                        mf.vm_need = ("S", spos[f])
                    else:
                        mf.lds_need = ("A", lpos[f])
                if m == first_of_tile[t] + X:
                    self.wait_lds(("ci", t, X))
                for n in range(4):
                    assert "v%d" % (bq + n) in written, ("B quad not written before MFMA", m, t, k, X, bq)
                P(mf)
                self.stats["mfma"] += 1
                used = 0
                if X == 0:
                    # refill the slot the PREVIOUS fragment's MFMAs consumed (they issued before this one)
                    if k[0] != "S":
                        i = lpos[f]
                        if i >= 1 and next_l < len(lds_ops) and next_l <= i - 1 + RING_L:
                            ins = lds_read(next_l)
                            P(ins)
                            used += ins.cost
                            next_l += 1
                    while next_s < len(s_ops) and used + 7 <= self.budget + 6:
                        if next_s >= RING_S and not s_ops[next_s - RING_S] < f:
                            break
                        for ins in s_req(next_s):
                            P(ins)
                            used += ins.cost
                        next_s += 1
                        if f < 12:
                            break
                while ci_pending and first_of_tile[ci_pending[0][0]] <= m + 24:
                    ins = ci_read(*ci_pending.pop(0))
                    P(ins)
                    used += ins.cost
                # ---- riders
                if not stream_done(0):
                    active = [0]
                else:
                    active = [si for si in active if si != 0 and not stream_done(si)]
                    for si in range(1, len(streams)):
                        if len(active) >= 2:
                            break
                        if si in active or stream_done(si):
                            continue
                        if active and (si % 2) == (active[0] % 2):
                            continue
                        if any(not stream_done(sj) and sj not in active and (sj % 2) == (si % 2) for sj in range(1, si)):
                            continue
                        active.append(si)
                progress = True
                while used < self.budget and progress:
                    progress = False
                    for si in list(active):
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
        left = []
        for si in range(len(streams)):
            left += streams[si][heads[si]:]
        assert not ci_pending and next_l == len(lds_ops) and next_s == len(s_ops), (ci_pending, next_l, len(lds_ops), next_s, len(s_ops))
        for ins in left:
            P(ins)
        pad = 16 - len(left)
        while pad > 0:
            self.raw("s_nop %d" % (min(pad, 8) - 1), 4 * min(pad, 8), "nop")
            pad -= 8
        for ins in rem_q:
            P(ins)
        self.raw("s_waitcnt lgkmcnt(0)", 1, "wait")
        self.stats["tail_left"] = len(left)
        return self.out


HEADER = '''// GENERATED by tools/gen_riders16_asm.py - do not edit (edit the generator).
// The whole wave-step of the bf16x3 flip pass at 69..100 units on v_mfma_f32_16x16x32_bf16 (split16_core.h: S16Layout) as one
// hand-scheduled asm block: 722 MFMAs of 16 cycles, two chain sets per wave, every A fragment read once for both.
#pragma once
#include "split16_core.h"

namespace rnnwf {

template <int NOUT> struct Riders16StepAsm { static constexpr bool kAvailable = false; };

'''


def struct_text(order23, budget):
    st = Step16(order23, budget)
    st.build()
    L = st.L
    clob = ["v%d" % n for n in list(range(48, 64)) + list(range(128, 256))] + ["a%d" % n for n in range(256)] + \
           ["s40", "s41", "s42", "s43", "s44", "memory"]
    clobs = ", ".join('"%s"' % c for c in clob)
    body = "\\n\\t".join(i.text for i in st.out if i.text)
    n_ins = sum(1 for i in st.out if i.text)
    issue = sum(i.cost for i in st.out)
    tmpl = '''// @NINS@ instructions, @NMFMA@ MFMAs; issue-cost model: @ISSUE@ cycles per wave-step (matrix pipe: @PIPE@); riders left for the tail: @LEFT@
template <> struct Riders16StepAsm<1> {
    static constexpr bool kAvailable = true;
    using L = S16Layout<1>;
    static_assert(L::LDS_REG == @LDSREG@ && L::OFF_CI - L::LSHIFT == @CI@ && L::OFF_XC - L::LSHIFT == @XC@ && L::OFF_ASP == @ASP@ && L::NT == 19 && L::XCP == 28,
                  "generated for another layout: re-run tools/gen_riders16_asm.py");
    // h0..h3: this lane's 2 x 25 state values (set X, unit 4 j + g at entry 25 X + j of the 64) in, the new state out; entries 50 + X out:
    // the head logit of the state that ENTERED the step, per set.  ci0 / ci1, xc0 / xc1: LDS byte addresses of the lane group's
    // accumulator-table row of tile 0 and of its candidate-input row for the input spin of set 0 / 1.  l0..l2: LDS byte address of
    // this lane's 16 bytes of fragment 0, + 0 / 64 / 128 KB.  vo: lane * 16.  rs: buffer descriptor of the image's fragment area.
    static __device__ __forceinline__ void run(f32x16& h0, f32x16& h1, f32x16& h2, f32x16& h3, unsigned ci0, unsigned ci1, unsigned xc0,
                                               unsigned xc1, unsigned l0, unsigned l1, unsigned l2, unsigned vo, u32x4 rs) {
        asm volatile("@BODY@"
                     : "+{v[64:79]}"(h0), "+{v[80:95]}"(h1), "+{v[96:111]}"(h2), "+{v[112:127]}"(h3)
                     : [ci0] "v"(ci0), [ci1] "v"(ci1), [xc0] "v"(xc0), [xc1] "v"(xc1), [l0] "v"(l0), [l1] "v"(l1), [l2] "v"(l2),
                       [vo] "v"(vo), [rs] "s"(rs)
                     : @CLOB@);
    }
};
'''
    rep = {"@NINS@": n_ins, "@NMFMA@": st.stats["mfma"], "@ISSUE@": issue, "@PIPE@": 16 * st.stats["mfma"], "@LEFT@": st.stats["tail_left"],
           "@LDSREG@": LDS_REG, "@CI@": L["OFF_CI"] - L["LSHIFT"], "@XC@": L["OFF_XC"] - L["LSHIFT"], "@ASP@": L["OFF_ASP"],
           "@BODY@": body.replace("%%", "%"), "@CLOB@": clobs}
    for k, v in rep.items():
        tmpl = tmpl.replace(k, str(v))
    return tmpl, st


def main():
    order23 = os.environ.get("RIDERS_ORDER", "tile")
    budget = int(os.environ.get("RIDERS_BUDGET", "8"))
    txt, st = struct_text(order23, budget)
    print("order=%s budget=%d:" % (order23, budget), st.stats, "instructions", sum(1 for i in st.out if i.text), "issue cycles", sum(i.cost for i in st.out))
    out = HEADER + txt + "\n}  // namespace rnnwf\n"
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "rnnwavefunctions_amd", "csrc", "split_riders16_asm.h")
    with open(path, "w") as f:
        f.write(out)
    print("wrote", path)


if __name__ == "__main__":
    main()
