#!/usr/bin/env python3
"""Generates rnnwavefunctions_amd/csrc/split_mfma_asm.h: the MFMA segment of the bf16x3 GRU step as ONE hand-scheduled
inline-asm block per K-packed layout (split_core.h, MODE 2).

Why by hand: the segment is 95 v_mfma_f32_32x32x16_bf16 fed by 95 ds_read_b128 of weight fragments.  hipcc, short of
registers, sinks every fragment read to just in front of the MFMA that consumes it, so each MFMA waits out an LDS
round trip (in-kernel stamps, tools/stamps.py: 51 cycles per MFMA instead of 32).  Here the reads run TWO k-steps
(9-10 MFMAs, ~300 cycles) ahead through a ring of two fragment sets, every wait is a counted `s_waitcnt lgkmcnt(8)`,
and nothing but MFMAs, reads and waits is issued, which leaves the SIMD's vector issue to the partner wave's VALU
segment (prnn_flip_pp_kernel).

Schedule, MFMA m = (k-step k, tile t):   wait until fragment m has landed;  MFMA m;  read fragment m - 1 + 2 NT into the
registers MFMA m - 1 consumed.
"""
import os
import sys

ORD = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]       # (weight part, state part), smallest products first


def gen_body(NT, NQ, zero=()):
    """(declarations, asm body, outputs, inputs) of one block; tiles in `zero` start from 0 (first MFMA with the inline constant
    as C, early-clobber output) instead of accumulating on their incoming value."""
    KS = 6 * NQ + 1
    NB = 3 * NQ + 1

    def a_off(t, k):
        if k < 6 * NQ:
            return ((t * 3 + ORD[k // NQ][0]) * NQ + k % NQ) * 1024
        return (NT * 3 * NQ + t) * 1024

    def b_idx(k):
        return ORD[k // NQ][1] * NQ + k % NQ if k < 6 * NQ else 3 * NQ

    total = KS * NT
    RING = 2                                   # fragment sets; set k % RING holds k-step k
    lines = []
    for k in range(min(RING, KS)):
        for t in range(NT):
            lines.append("ds_read_b128 %%[f%d_%d], %%[aa] offset:%d" % (k % RING, t, a_off(t, k)))
    issued = min(RING, KS) * NT

    def read(r):                               # read of fragment index r = k NT + t
        k, t = divmod(r, NT)
        return "ds_read_b128 %%[f%d_%d], %%[aa] offset:%d" % (k % RING, t, a_off(t, k))

    for m in range(total):
        k, t = divmod(m, NT)
        lines.append("s_waitcnt lgkmcnt(%d)" % (issued - m - 1))
        if k == 0 and t in zero:
            lines.append("v_mfma_f32_32x32x16_bf16 %%[c%d], %%[f%d_%d], %%[b%d], 0" % (t, k % RING, t, b_idx(k)))
        else:
            lines.append("v_mfma_f32_32x32x16_bf16 %%[c%d], %%[f%d_%d], %%[b%d], %%[c%d]" % (t, k % RING, t, b_idx(k), t))
        # the register set of fragment m - 1 is free once MFMA m has issued (MFMA m - 1 started >= 32 cycles ago and
        # has long read its operands): refill it with the fragment RING k-steps further on
        r = m - 1 + RING * NT
        if m >= 1 and r < total:
            lines.append(read(r))
            issued += 1
    r = total - 1 + RING * NT
    assert issued == total, (issued, total)
    # the accumulators are read by VALU code right behind this block (only a barrier in between): an 8-pass MFMA's
    # result needs >= 10 wait states before a VALU read, and the assembler inserts none inside inline asm
    lines += ["s_nop 7", "s_nop 7"]
    body = "\\n\\t".join(lines).replace("%%", "%")
    ins = ", ".join(['[b%d] "v"(B[%d])' % (i, i) for i in range(NB)] + ['[aa] "v"(a_addr)'])
    decl = "u32x4 " + ", ".join("f%d_%d" % (r, t) for r in range(RING) for t in range(NT)) + ";"
    outs = ", ".join(['[c%d] "%s"(c%d)' % (t, "=&v" if t in zero else "+v", t) for t in range(NT)] +
                     ['[f%d_%d] "=&v"(f%d_%d)' % (r, t, r, t) for r in range(RING) for t in range(NT)])
    return decl, body, outs, ins


def gen(NT, NQ):
    NB = 3 * NQ + 1
    cargs = ", ".join("f32x16& c%d" % t for t in range(NT))
    ccall = ", ".join("acc[%d]" % t for t in range(NT))
    fn = '''    static __device__ __forceinline__ void %s(unsigned a_addr, const u32x4 (&B)[%d], %s) {
        %s
        asm volatile("%s"
                     : %s
                     : %s
                     : "memory");
    }
'''
    variants = ""
    for name, zero in (("run_tiles", ()), ("run_tiles_from_zero", tuple(range(NT))), ("run_tiles_zero_2_4", (2, 4))):
        decl, body, outs, ins = gen_body(NT, NQ, zero)
        variants += fn % (name, NB, cargs, decl, body, outs, ins)
    return '''template <> struct MfmaSegAsm<%d, %d> {
    static constexpr bool kAvailable = true;
    // a_addr: LDS byte address of this lane's 16 bytes in fragment 0 of the A image; B: the %d state quads
    // (3 parts x %d k-steps, then the special k-step); c0..: in = bias / one-hot rows (or the sums so far), out = pre-activations.
    // Separate accumulator references: the stacked-layer kernels run the block twice per step - X block, H block - on
    // different subsets of their seven accumulator tiles (split_pp.h: SplitPPUpper); their biases travel in two spare K entries of
    // the special k-step, so run_tiles_from_zero starts every tile from 0 (no accumulator preload, no registers held for it) and
    // run_tiles_zero_2_4 continues tiles 0, 1, 3 (r, u, mixed 0) while tiles 2, 4 (q, mixed 1 of the H block) start from 0.
%s    static __device__ __forceinline__ void run(unsigned a_addr, const u32x4 (&B)[%d], f32x16 (&acc)[%d]) {
        run_tiles(a_addr, B, %s);
    }
};
''' % (NT, NQ, NB, NQ, variants, NB, NT, ccall)


HEADER = '''// GENERATED by tools/gen_split_mfma_asm.py - do not edit (edit the generator).
// MFMA segment of the bf16x3 GRU step, hand-scheduled: fragment reads two k-steps ahead of their MFMAs, counted waits.
#pragma once

namespace rnnwf {

template <int NT, int NQ> struct MfmaSegAsm { static constexpr bool kAvailable = false; };

'''


def main():
    out = HEADER
    for NT, NQ in ((5, 3),):          # MODE 2 at 37..50 units: NT = 3 NF32 + NMIX = 5 tiles, NQ = 3 k-steps per part
        out += gen(NT, NQ) + "\n"
    out += "}  // namespace rnnwf\n"
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "rnnwavefunctions_amd", "csrc", "split_mfma_asm.h")
    with open(path, "w") as f:
        f.write(out)
    print("wrote", path)


if __name__ == "__main__":
    main()
