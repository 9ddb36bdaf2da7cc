#!/usr/bin/env python3
"""Generates rnnwavefunctions_amd/csrc/split_riders_asm.h: ONE hand-scheduled inline-asm block for the whole wave-step of
the bf16x3 "riders" kernel at 69..100 units (split_core.h, SplitLayout<3, 2, NOUT, 3>, streamed w3 fragments; config 5).

Why by hand: a wave-step is 380 v_mfma_f32_32x32x16_bf16 with ~1 000 VALU / transcendental instructions, 380 fragment
reads (LDS) or loads (L2) and their waits riding between them.  One wave per SIMD leaves 32 - 8 = 24 issue cycles per
MFMA for everything else; the step needs ~19 of them on average, so it FITS - but only if the riders are spread evenly
and never wait on each other.  hipcc's order (split_core.h: step_stream) runs at 46 cycles per MFMA; the bound is 32.

What this generator does (everything is static, decided here, in Python):
  * fixes the MFMA order (pass 1 k-major over tiles 0..2 - the state parts h2 / h3 appear while it runs - then tiles 3..5,
    6..8, and the remainder tile 9 last) and every register by hand (VGPR / AGPR map below);
  * keeps A fragments in two AGPR rings - LDS-fed ones ~9 MFMAs ahead, streamed ones ~12 quads (> 1 000 cycles) ahead,
    all of a step's streamed requests issued inside the step (nothing carried between steps) - with counted waits;
  * lays every VALU rider out stage by stage over batches of eight units (all exp2, all +1, all rcp ...) and list-
    schedules the rider streams into the slots behind the MFMAs under an issue-cycle budget, subject to readiness
    (a gate may read an accumulator tile only two MFMAs after the tile's last MFMA; h2 / h3 quads before their first
    use), inserting the wait states the hardware does not interlock (VALU -> MFMA operand 2, transcendental -> VALU 1,
    MFMA -> accvgpr_read 16).

Register map (the kernel around the block keeps v0..v47 for itself; h travels as four pinned f32x16 operands):
  v48..v63    temporaries (split) / gate batch B (gc, gx)          v64..v113   h[50] (in / out), v114 head logit (out)
  v116..v127  remainder units' gate temporaries                    v128..v199  R[3][24] state parts (MFMA B quads)
  v200..v207  RS[2][4] special k-steps' B quads                    v208..v255  split residuals (pass 1) / gate batches A, B
  a0..a159    accumulators of tiles 0..9                           a160..a199  LDS fragment ring (10 quads)
  a200..a255  streamed fragment ring (14 quads)
"""
import os
import sys

NF32, RJ, NT, NQ, KSP, NU, NUA, NS = 3, 2, 10, 6, 2, 50, 48, 2
NRM = 24
HEAD_SLOT = 3 * RJ
LDS_REG = NT * 2 * NQ * 1024               # resident regular fragments (parts 0, 1), then the special fragments
RING_L, RING_S = 10, 14
MFMA_ISSUE, SLOT = 8, 32


def layout(nout):
    off_asp = NT * 3 * NQ * 1024
    sz_a = NT * (3 * NQ + KSP) * 1024
    off_ci = sz_a
    off_xc = off_ci + 2 * NT * 2 * 16 * 4
    nup = ((NU + 3) // 4) * 4
    off_wd = off_xc + 2 * 2 * nup * 4
    off_bd = off_wd + 2 * nup * nout * 4
    total = off_bd + 16
    lshift = off_asp - LDS_REG
    return dict(OFF_ASP=off_asp, OFF_CI=off_ci, OFF_XC=off_xc, BYTES=total, LSHIFT=lshift, LDS_BYTES=total - lshift, NUP=nup)


def vh(e): return 64 + e
def vR(p, i): return 128 + 24 * p + i
def vRS(k, j): return 200 + 4 * k + j
def aacc(t, s): return 16 * t + s
def aringL(r): return 160 + 4 * r
def aringS(r): return 200 + 4 * r


ORD5 = [(1, 0), (0, 0), (1, 1), (0, 1), (0, 2)]           # LDS-fed products (weight part, state part), streamed layout


class Ins:
    __slots__ = ("text", "cost", "reads", "writes", "kind", "ready", "lds_need", "lds_tag", "vm_need", "vm_tag", "accread")

    def __init__(self, text, cost, reads=(), writes=(), kind="valu", ready=-1):
        self.text, self.cost, self.reads, self.writes, self.kind, self.ready = text, cost, set(reads), set(writes), kind, ready
        self.lds_need = self.lds_tag = self.vm_need = self.vm_tag = None
        self.accread = False


def V(n): return "v%d" % n
def A(n): return "a%d" % n
def VQ(n): return "v[%d:%d]" % (n, n + 3)
def AQ(n): return "a[%d:%d]" % (n, n + 3)


def valu(op, dst, *src, cost=4, kind="valu", ready=-1):
    regs = [s for s in src if isinstance(s, int)]
    txt = "%s %s, %s" % (op, V(dst), ", ".join(V(s) if isinstance(s, int) else s for s in src))
    return Ins(txt, cost, ["v%d" % r for r in regs], ["v%d" % dst], kind, ready)


def trans(op, dst, src, ready=-1):
    return valu(op, dst, src, cost=8, kind="trans", ready=ready)


class Step:
    def __init__(self, nout, order23="tile", budget=22):
        self.nout, self.budget, self.order23 = nout, budget, order23
        self.L = layout(nout)
        self.out = []                      # emitted Ins
        self.lds_q = []                    # outstanding LDS op tags, in issue order
        self.vm_q = []
        self.stats = {"mfma": 0, "riders": 0, "nops": 0, "waits": 0, "over": 0}

    # ---- emission with hazard / wait handling -------------------------------------------------------------------------
    def emit(self, ins):
        prev = self.out[-1] if self.out else None
        if ins.lds_need is not None and ins.lds_need in self.lds_q:
            newer = len(self.lds_q) - 1 - self.lds_q.index(ins.lds_need)
            self.raw("s_waitcnt lgkmcnt(%d)" % min(newer, 15), 1, "wait")
            keep = min(newer, 15)
            self.lds_q = self.lds_q[len(self.lds_q) - keep:] if keep else []
            prev = self.out[-1]
        if ins.vm_need is not None and ins.vm_need in self.vm_q:
            newer = len(self.vm_q) - 1 - self.vm_q.index(ins.vm_need)
            self.raw("s_waitcnt vmcnt(%d)" % min(newer, 63), 1, "wait")
            keep = min(newer, 63)
            self.vm_q = self.vm_q[len(self.vm_q) - keep:] if keep else []
            prev = self.out[-1]
        # transcendental result -> next VALU: one wait state
        if prev is not None and prev.kind == "trans" and ins.kind in ("valu", "trans", "mfma") and (prev.writes & ins.reads):
            self.raw("s_nop 0", 4, "nop")
        # VALU write -> MFMA A / B operand: two wait states
        if ins.kind == "mfma":
            need = 0
            for back, p in enumerate(reversed(self.out[-2:])):
                if p.kind in ("valu", "trans") and (p.writes & ins.reads):
                    need = max(need, 2 - back)
            if need:
                self.raw("s_nop %d" % (need - 1), 4 * need, "nop")
        self.out.append(ins)
        if ins.lds_tag is not None:
            self.lds_q.append(ins.lds_tag)
        if ins.vm_tag is not None:
            self.vm_q.append(ins.vm_tag)

    def wait_lds(self, tag):
        if tag in self.lds_q:
            newer = len(self.lds_q) - 1 - self.lds_q.index(tag)
            self.raw("s_waitcnt lgkmcnt(%d)" % min(newer, 15), 1, "wait")
            keep = min(newer, 15)
            self.lds_q = self.lds_q[len(self.lds_q) - keep:] if keep else []

    def raw(self, text, cost, kind):
        self.out.append(Ins(text, cost, kind=kind))
        self.stats["nops" if kind == "nop" else "waits"] += 1

    # ---- operand addressing -------------------------------------------------------------------------------------------
    def lds_frag_addr(self, t, kind, idx):
        if kind == "L":
            wp = ORD5[idx // NQ][0]
            off = ((t * 2 + wp) * NQ + idx % NQ) * 1024
        else:
            off = LDS_REG + (t * KSP + idx) * 1024
        return "%%[l%d] offset:%d" % (off >> 16, off & 0xffff)

    # ---- the MFMA list -----------------------------------------------------------------------------------------------
    def mfma_list(self):
        std = []
        for j in range(NQ):
            std.append(("S", j))
            std += [("L", 5 * j + i) for i in range(5)]
        std += [("P", 0), ("P", 1)]
        p1 = [("L", k) for k in range(2 * NQ)]
        for j in range(NQ):
            p1.append(("S", j))
            p1 += [("L", 2 * NQ + 3 * j + i) for i in range(3)]
        p1 += [("P", 0), ("P", 1)]
        assert len(std) == 38 and len(p1) == 38 and sorted(x for x in p1 if x[0] == "L") == sorted(x for x in std if x[0] == "L")
        seq = [(t, k) for k in p1 for t in (0, 1, 2)]
        for base in (3, 6):
            if self.order23 == "tile":
                seq += [(t, k) for t in range(base, base + 3) for k in std]
            else:
                seq += [(t, k) for k in std for t in range(base, base + 3)]
        seq += [(9, k) for k in std]
        return seq

    def b_quad(self, k):
        kind, idx = k
        if kind == "L":
            return vR(ORD5[idx // NQ][1], 4 * (idx % NQ))
        if kind == "S":
            return vR(0, 4 * idx)
        return vRS(idx, 0)

    # ---- rider streams -----------------------------------------------------------------------------------------------
    def split_riders(self):
        """h1 quads 1..5 (quad 0 is converted in front of the first MFMA), then h2, h3 and the special units' quads."""
        q = []
        T = list(range(48, 64))
        for x in range(1, NQ):
            for i in range(4 * x, 4 * x + 4):
                q.append(valu("v_cvt_pk_bf16_f32", vR(0, i), vh(2 * i), vh(2 * i + 1)))
        res0 = lambda i: 208 + i
        res1 = lambda i: 232 + i
        for x in range(NQ):                                    # h2 of quad x, stage by stage over its four pairs
            P = range(4 * x, 4 * x + 4)
            for n, i in enumerate(P): q.append(valu("v_lshlrev_b32", T[n], "16", vR(0, i)))
            for n, i in enumerate(P): q.append(valu("v_and_b32", T[4 + n], "0xffff0000", vR(0, i)))
            for n, i in enumerate(P): q.append(valu("v_sub_f32", res0(i), vh(2 * i), T[n]))
            for n, i in enumerate(P): q.append(valu("v_sub_f32", res1(i), vh(2 * i + 1), T[4 + n]))
            for n, i in enumerate(P): q.append(valu("v_cvt_pk_bf16_f32", vR(1, i), res0(i), res1(i)))
        for x in range(NQ):                                    # h3 of quad x
            P = range(4 * x, 4 * x + 4)
            for n, i in enumerate(P): q.append(valu("v_lshlrev_b32", T[8 + n], "16", vR(1, i)))
            for n, i in enumerate(P): q.append(valu("v_and_b32", T[12 + n], "0xffff0000", vR(1, i)))
            for n, i in enumerate(P): q.append(valu("v_sub_f32", T[8 + n], res0(i), T[8 + n]))
            for n, i in enumerate(P): q.append(valu("v_sub_f32", T[12 + n], res1(i), T[12 + n]))
            for n, i in enumerate(P): q.append(valu("v_cvt_pk_bf16_f32", vR(2, i), T[8 + n], T[12 + n]))
        # special units: q[sp][part] with the part in both halves of the register
        Q = [[T[0], T[1], T[2]], [T[3], T[4], T[5]]]
        X = [vh(NUA), vh(NUA + 1)]
        for sp in range(NS): q.append(valu("v_cvt_pk_bf16_f32", Q[sp][0], X[sp], X[sp]))
        for sp in range(NS): q.append(valu("v_lshlrev_b32", T[6 + sp], "16", Q[sp][0]))
        for sp in range(NS): q.append(valu("v_sub_f32", T[6 + sp], X[sp], T[6 + sp]))
        for sp in range(NS): q.append(valu("v_cvt_pk_bf16_f32", Q[sp][1], T[6 + sp], T[6 + sp]))
        for sp in range(NS): q.append(valu("v_lshlrev_b32", T[8 + sp], "16", Q[sp][1]))
        for sp in range(NS): q.append(valu("v_sub_f32", T[8 + sp], T[6 + sp], T[8 + sp]))
        for sp in range(NS): q.append(valu("v_cvt_pk_bf16_f32", Q[sp][2], T[8 + sp], T[8 + sp]))
        PH = [0, 1, 2, 0, 1, 0]
        for k in range(KSP):
            for j in range(4):
                e0, e1 = 8 * k + 2 * j, 8 * k + 2 * j + 1
                lo = Q[e0 // 6][PH[e0 % 6]] if e0 < 6 * NS else None
                hi = Q[e1 // 6][PH[e1 % 6]] if e1 < 6 * NS else None
                if lo is None and hi is None:
                    q.append(valu("v_mov_b32", vRS(k, j), "0"))
                else:
                    assert lo is not None and hi is not None
                    ins = valu("v_bfi_b32", vRS(k, j), "s44", lo, hi)      # (0xffff & lo) | (~0xffff & hi)
                    q.append(ins)
        return q

    def gate_batch(self, b, hb, regs, tile_done):
        """Gates of entries 16 b + 8 hb + u, u < 8: three parts, each ready two MFMAs after its tile's last MFMA."""
        gr, gu, gc, gx = regs
        e0 = 16 * b + 8 * hb
        U = range(8)
        rdy = lambda t: tile_done[t] + 3

        def part(tile, g):
            q = []
            for u in U:
                i = Ins("v_accvgpr_read_b32 %s, %s" % (V(g[u]), A(aacc(tile, 8 * hb + u))), 4, ["a%d" % aacc(tile, 8 * hb + u)], ["v%d" % g[u]], "valu", rdy(tile))
                i.accread = True
                q.append(i)
            return q
        qr = part(3 * b, gr)
        qr += [trans("v_exp_f32", gr[u], gr[u]) for u in U]
        qr += [valu("v_add_f32", gr[u], "1.0", gr[u]) for u in U]
        qr += [trans("v_rcp_f32", gr[u], gr[u]) for u in U]
        qu = part(3 * b + 1, gu)
        qu += [trans("v_exp_f32", gu[u], gu[u]) for u in U]
        qu += [valu("v_add_f32", gu[u], "1.0", gu[u]) for u in U]
        qu += [trans("v_rcp_f32", gu[u], gu[u]) for u in U]
        qc = part(3 * b + 2, gc)
        for w in range(2):
            i = Ins("ds_read_b128 %s, %%[xc] offset:%d" % (VQ(gx[4 * w]), (e0 + 4 * w) * 4), 2, (), ["v%d" % (gx[4 * w] + n) for n in range(4)], "lds")
            i.lds_tag = ("xc", b, hb, w)
            qc.append(i)
        for u in U:
            i = valu("v_fma_f32", gc[u], gr[u], gc[u], gx[u])
            i.lds_need = ("xc", b, hb, u // 4)
            qc.append(i)
        qc += [trans("v_exp_f32", gc[u], gc[u]) for u in U]
        qc += [valu("v_add_f32", gc[u], "1.0", gc[u]) for u in U]
        qc += [trans("v_rcp_f32", gc[u], gc[u]) for u in U]
        qc += [valu("v_fma_f32", gc[u], "2.0", gc[u], "-1.0") for u in U]
        qc += [valu("v_sub_f32", gr[u], vh(e0 + u), gc[u]) for u in U]
        qc += [valu("v_fma_f32", vh(e0 + u), gu[u], gr[u], gc[u]) for u in U]
        return qr + qu + qc

    def remainder_gates(self):
        """Entries 48, 49 from tile 9 (slots r: 0, 1; u: 2, 3; candidate: 4, 5) and the head logit (slot 6): the step's tail."""
        gr, gu, gc, gx = [118, 119], [120, 121], [122, 123], [116, 117]
        q = []
        i = Ins("ds_read_b64 v[116:117], %%[xc] offset:%d" % (NUA * 4), 2, (), ["v116", "v117"], "lds")
        i.lds_tag = ("xc", "rem")
        q.append(i)
        for g, s0 in ((gr, 0), (gu, RJ), (gc, 2 * RJ)):
            for j in range(RJ):
                r = Ins("v_accvgpr_read_b32 %s, %s" % (V(g[j]), A(aacc(9, s0 + j))), 4, ["a%d" % aacc(9, s0 + j)], ["v%d" % g[j]])
                r.accread = True
                q.append(r)
        for o in range(self.nout):
            r = Ins("v_accvgpr_read_b32 %s, %s" % (V(114 + o), A(aacc(9, HEAD_SLOT + o))), 4, ["a%d" % aacc(9, HEAD_SLOT + o)], ["v%d" % (114 + o)])
            r.accread = True
            q.append(r)
        J = range(RJ)
        q += [trans("v_exp_f32", g[j], g[j]) for g in (gr, gu) for j in J]
        q += [valu("v_add_f32", g[j], "1.0", g[j]) for g in (gr, gu) for j in J]
        q += [trans("v_rcp_f32", g[j], g[j]) for g in (gr, gu) for j in J]
        for j in J:
            f = valu("v_fma_f32", gc[j], gr[j], gc[j], gx[j])
            f.lds_need = ("xc", "rem")
            q.append(f)
        q += [trans("v_exp_f32", gc[j], gc[j]) for j in J]
        q += [valu("v_add_f32", gc[j], "1.0", gc[j]) for j in J]
        q += [trans("v_rcp_f32", gc[j], gc[j]) for j in J]
        q += [valu("v_fma_f32", gc[j], "2.0", gc[j], "-1.0") for j in J]
        q += [valu("v_sub_f32", gr[j], vh(NUA + j), gc[j]) for j in J]
        q += [valu("v_fma_f32", vh(NUA + j), gu[j], gr[j], gc[j]) for j in J]
        return q

    # ---- the schedule ------------------------------------------------------------------------------------------------
    def build(self):
        seq = self.mfma_list()
        nm = len(seq)
        tile_done = {}
        for m, (t, k) in enumerate(seq):
            tile_done[t] = m
        first_of_tile = {}
        for m, (t, k) in enumerate(seq):
            first_of_tile.setdefault(t, m)
        lds_ops = [m for m, (t, k) in enumerate(seq) if k[0] != "S"]          # MFMAs with an LDS-fed A fragment, in order
        s_ops = [m for m, (t, k) in enumerate(seq) if k[0] == "S"]
        lpos = {m: i for i, m in enumerate(lds_ops)}
        spos = {m: i for i, m in enumerate(s_ops)}

        def lds_read(i):
            m = lds_ops[i]
            t, k = seq[m]
            ins = Ins("ds_read_b128 %s, %s" % (AQ(aringL(i % RING_L)), self.lds_frag_addr(t, k[0], k[1])), 2, (),
                      ["a%d" % (aringL(i % RING_L) + n) for n in range(4)], "lds")
            ins.lds_tag = ("A", i)
            return ins

        def s_req(i):
            m = s_ops[i]
            t, k = seq[m]
            soff = ((t * 3 + 2) * NQ + k[1]) * 1024
            sreg = "s%d" % (40 + i % 4)
            mov = Ins("s_mov_b32 %s, 0x%x" % (sreg, soff), 1, kind="salu")
            ld = Ins("buffer_load_dwordx4 %s, %%[vo], %%[rs], %s offen" % (AQ(aringS(i % RING_S)), sreg), 6, (),
                     ["a%d" % (aringS(i % RING_S) + n) for n in range(4)], "vmem")
            ld.vm_tag = ("S", i)
            return [mov, ld]

        # rider streams: split (pass 1), then the gate batches in block order; two register sets alternate
        regsA = (list(range(208, 216)), list(range(216, 224)), list(range(224, 232)), list(range(232, 240)))
        regsB = (list(range(240, 248)), list(range(248, 256)), list(range(48, 56)), list(range(56, 64)))
        split_q = self.split_riders()
        last_split_mfma = max(m for m, (t, k) in enumerate(seq) if t <= 2)           # pass 1 ends here
        batches = []
        for b in range(NF32):
            for hb in range(2):
                regs = regsA if (2 * b + hb) % 2 == 0 else regsB
                q = self.gate_batch(b, hb, regs, tile_done)
                for ins in q:
                    # batch registers overlap the split's residuals / temporaries: no gate before pass 1's last MFMA
                    ins.ready = max(ins.ready, last_split_mfma + 1)
                batches.append(q)
        rem_q = self.remainder_gates()

        # ---- prologue: scalar constants, accumulator tables of every tile that starts early, first fragments, h1 quad 0
        P = self.emit
        self.raw("s_mov_b32 s44, 0xffff", 1, "salu")
        self.raw("s_waitcnt lgkmcnt(0)", 1, "wait")

        def ci_reads(t):
            out = []
            for n in range(4):
                ins = Ins("ds_read_b128 %s, %%[ci] offset:%d" % (AQ(aacc(t, 4 * n)), t * 128 + 16 * n), 2, (),
                          ["a%d" % aacc(t, 4 * n + x) for x in range(4)], "lds")
                ins.lds_tag = ("ci", t, n)
                out.append(ins)
            return out
        for t in (0, 1, 2):
            for ins in ci_reads(t):
                P(ins)
        n_l0 = min(RING_L - 1, len(lds_ops))
        for i in range(n_l0):
            P(lds_read(i))
        next_l = n_l0
        for i in range(4):
            P(valu("v_cvt_pk_bf16_f32", vR(0, i), vh(2 * i), vh(2 * i + 1)))
        next_s = 0
        ci_pending = [t for t in range(3, NT)]                 # tables of the later tiles: issued as riders ahead of their first MFMA
        written = set("v%d" % vR(0, i) for i in range(4))

        streams = [split_q] + batches
        heads = [0] * len(streams)
        active = [0]                                           # indices of the streams riders are drawn from (at most two gate batches)

        def stream_done(si): return heads[si] >= len(streams[si])

        for m, (t, k) in enumerate(seq):
            # ---- the MFMA and the wait for its A fragment
            bq = self.b_quad(k)
            if k[0] == "S":
                areg = aringS(spos[m] % RING_S)
            else:
                areg = aringL(lpos[m] % RING_L)
            mf = Ins("v_mfma_f32_32x32x16_bf16 a[%d:%d], %s, %s, a[%d:%d]" % (aacc(t, 0), aacc(t, 15), AQ(areg), VQ(bq), aacc(t, 0), aacc(t, 15)),
                     MFMA_ISSUE, ["v%d" % (bq + n) for n in range(4)], (), "mfma")
            if k[0] == "S":
                mf.vm_need = ("S", spos[m])
            else:
                mf.lds_need = ("A", lpos[m])
            if m == first_of_tile[t]:
                self.wait_lds(("ci", t, 3))                    # the tile's accumulator table (four reads, in order) has landed
            for n in range(4):
                assert "v%d" % (bq + n) in written, ("B quad not written before MFMA", m, t, k, bq)
            P(mf)
            self.stats["mfma"] += 1
            used = 0
            # ---- operand traffic: refill the fragment slot MFMA m - 1 consumed; requests of streamed quads
            if k[0] != "S":
                i = lpos[m]
                if i >= 1 and next_l < len(lds_ops) and next_l <= i - 1 + RING_L:
                    ins = lds_read(next_l)
                    P(ins)
                    used += ins.cost
                    next_l += 1
            # streamed ring: the slot of request j is free once the MFMA that consumed request j - RING_S has issued and one
            # more MFMA (this one) behind it
            while next_s < len(s_ops) and used + 7 <= self.budget:
                if next_s >= RING_S and not s_ops[next_s - RING_S] < m:
                    break
                for ins in s_req(next_s):
                    P(ins)
                    used += ins.cost
                next_s += 1
                if m < 12:                                     # the first requests: one per slot (pass 1 starts on LDS-fed products)
                    break
            # accumulator tables of tiles that start within the next few MFMAs
            while ci_pending and first_of_tile[ci_pending[0]] <= m + 12:
                for ins in ci_reads(ci_pending.pop(0)):
                    P(ins)
                    used += ins.cost
            # ---- riders
            if not stream_done(0):
                active = [0]
            else:
                active = [si for si in active if si != 0 and not stream_done(si)]
                for si in range(1, len(streams)):              # batches in order; two in flight only with different register sets
                    if len(active) >= 2:
                        break
                    if si in active or stream_done(si):
                        continue
                    if active and (si % 2) == (active[0] % 2):
                        continue
                    if any(not stream_done(sj) and sj not in active and (sj % 2) == (si % 2) for sj in range(1, si)):
                        continue                               # an earlier batch on the same register set comes first
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
                    if used + ins.cost > self.budget + 2:
                        continue
                    n0 = len(self.out)
                    P(ins)
                    used += sum(x.cost for x in self.out[n0:])
                    written |= ins.writes
                    heads[si] += 1
                    self.stats["riders"] += 1
                    progress = True
            if used > self.budget + 4:
                self.stats["over"] += 1
        # ---- tail: whatever is left of the gate batches, then the remainder units (the last MFMA finished >= 16 wait states ago)
        left = []
        for si in range(len(streams)):
            left += streams[si][heads[si]:]
        assert not ci_pending and next_l == len(lds_ops) and next_s == len(s_ops), (ci_pending, next_l, len(lds_ops), next_s, len(s_ops))
        n_left = len(left)
        for ins in left:
            P(ins)
        pad = 16 - sum(1 for _ in left)
        while pad > 0:
            self.raw("s_nop %d" % (min(pad, 8) - 1), 4 * min(pad, 8), "nop")
            pad -= 8
        for ins in rem_q:
            P(ins)
        assert not self.lds_q or all(x[0] != "A" for x in self.lds_q)
        self.raw("s_waitcnt lgkmcnt(0)", 1, "wait")
        self.stats["tail_left"] = n_left
        return self.out

    def text(self):
        return "\\n\\t".join(i.text for i in self.out if i.text)


HEADER = '''// GENERATED by tools/gen_riders_asm.py - do not edit (edit the generator).
// The whole wave-step of the bf16x3 riders kernel at 69..100 units (SplitLayout<3, 2, NOUT, 3>, streamed w3 fragments) as one
// hand-scheduled asm block: MFMA order, registers, fragment rings, counted waits and the placement of every VALU rider are fixed
// by the generator (see its docstring for the register map and the hazards it pads).
#pragma once
#include "split_core.h"

namespace rnnwf {

template <int NF32, int RJ, int NOUT, bool STREAM> struct RidersStepAsm { static constexpr bool kAvailable = false; };

'''


def struct_text(nout, order23, budget):
    st = Step(nout, order23, budget)
    st.build()
    L = st.L
    clob = ["v%d" % n for n in list(range(48, 64)) + list(range(128, 256))] + ["a%d" % n for n in range(256)] + \
           ["s40", "s41", "s42", "s43", "s44", "memory"]
    clobs = ", ".join('"%s"' % c for c in clob)
    body = st.text()
    n_ins = sum(1 for i in st.out if i.text)
    issue = sum(i.cost for i in st.out)
    tmpl = '''// @NINS@ instructions, @NMFMA@ MFMAs; issue-cost model: @ISSUE@ cycles per wave-step (matrix pipe: @PIPE@); riders left for the tail: @LEFT@
template <> struct RidersStepAsm<3, 2, @NOUT@, true> {
    static constexpr bool kAvailable = true;
    using L = SplitLayout<3, 2, @NOUT@, 3>;
    static_assert(L::STREAM && L::LDS_REG == @LDSREG@ && L::OFF_CI - L::LSHIFT == @CI@ && L::OFF_XC - L::LSHIFT == @XC@ && L::OFF_ASP == @ASP@ && L::NUP == @NUP@,
                  "generated for another layout: re-run tools/gen_riders_asm.py");
    // h0..h3: this lane's 50 state values (entry e at h[e / 16][e % 16]) in, the new state out; h3[2 + o] out: head row o of the
    // state that ENTERED the step.  ci / xc: LDS byte addresses of this lane's accumulator-table row of tile 0 and of its candidate
    // input row (both depend on the input spin).  l0..l2: LDS byte address of this lane's 16 bytes of fragment 0, + 0 / 64 / 128 KB.
    // vo: lane * 16.  rs: buffer descriptor of the global image's fragment area (w3 fragments are read from there).
    static __device__ __forceinline__ void run(f32x16& h0, f32x16& h1, f32x16& h2, f32x16& h3, unsigned ci, unsigned xc, unsigned l0,
                                               unsigned l1, unsigned l2, unsigned vo, u32x4 rs) {
        asm volatile("@BODY@"
                     : "+{v[64:79]}"(h0), "+{v[80:95]}"(h1), "+{v[96:111]}"(h2), "+{v[112:127]}"(h3)
                     : [ci] "v"(ci), [xc] "v"(xc), [l0] "v"(l0), [l1] "v"(l1), [l2] "v"(l2), [vo] "v"(vo), [rs] "s"(rs)
                     : @CLOB@);
    }
};
'''
    rep = {"@NINS@": n_ins, "@NMFMA@": st.stats["mfma"], "@ISSUE@": issue, "@PIPE@": 32 * st.stats["mfma"], "@LEFT@": st.stats["tail_left"],
           "@NOUT@": nout, "@LDSREG@": LDS_REG, "@CI@": L["OFF_CI"] - L["LSHIFT"], "@XC@": L["OFF_XC"] - L["LSHIFT"], "@ASP@": L["OFF_ASP"],
           "@NUP@": L["NUP"], "@BODY@": body.replace("%%", "%"), "@CLOB@": clobs}
    for k, v in rep.items():
        tmpl = tmpl.replace(k, str(v))
    return tmpl, st


def main():
    order23 = os.environ.get("RIDERS_ORDER", "tile")
    budget = int(os.environ.get("RIDERS_BUDGET", "22"))
    out = HEADER
    for nout in (1,):
        txt, st = struct_text(nout, order23, budget)
        out += txt + "\n"
        print("NOUT=%d order=%s budget=%d:" % (nout, order23, budget), st.stats, "instructions", sum(1 for i in st.out if i.text),
              "issue cycles", sum(i.cost for i in st.out))
    out += "}  // namespace rnnwf\n"
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "rnnwavefunctions_amd", "csrc", "split_riders_asm.h")
    with open(path, "w") as f:
        f.write(out)
    print("wrote", path)


if __name__ == "__main__":
    main()
