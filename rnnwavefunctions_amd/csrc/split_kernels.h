// split_kernels.h - flip pass of the positive RNN on the bf16x3 engine (split_core.h): same work items and
// outputs as prnn_flip_kernel (gru_kernels.h), 32 chains per wave.  Hidden-state checkpoints are read in the
// layout the f32 base pass wrote them ([N-1][ns/16][KT16][64], unit 4 kt + q of chain c16 at lane (q << 4) | c16).
#pragma once
#include "gru_kernels.h"
#include "split_core.h"
#include "split_pp.h"

namespace rnnwf {

template <int NF32, int RJ, int WAVES, int MODE>
__global__ void __launch_bounds__(WAVES * 64) prnn_flip_split_kernel(PrnnArgs a, const void* wsplit, int kt16) {
    using C = SplitCore<NF32, RJ, 1, MODE>;
    using L = typename C::L;
    constexpr int NU = C::NU, NR = C::NR;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, wsplit);
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int64_t nsb32 = (a.ns + 31) / 32;
    const int64_t ntiles = (int64_t)(N - 1) * nsb32;
    const float* hck = reinterpret_cast<const float*>(a.hck);
    for (int64_t tile = gw; tile < ntiles; tile += nw) {
        const int i = (int)(tile / nsb32);
        const int64_t sb = tile - (int64_t)i * nsb32;
        const int64_t s = sb * 32 + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        float h[NU];
        {
            const float* src = hck + (((int64_t)i * a.nsb + (sc >> 4)) * kt16) * 64 + (sc & 15);
#pragma unroll
            for (int e = 0; e < NU; ++e) {
                constexpr int dummy = 0; (void)dummy;
                const int u0 = L::unit_of(e, 0), u1 = L::unit_of(e, 1);
                const int u = hh ? u1 : u0;
                h[e] = u < 4 * kt16 ? src[(u >> 2) * 64 + ((u & 3) << 4)] : 0.0f;
            }
        }
        auto spin = [&](int n) { return (int)((a.bits[(int64_t)(n >> 5) * a.ns + sc] >> (n & 31)) & 1); };
        int sig_in = 1 - spin(i);
        double lp = 0.0;
        unsigned R[3][NR];
        u32x4 sf[2][L::RIDERS ? C::SFN : 1];
        if constexpr (L::STREAM) C::stream_first(C::stream_source(wsplit), sf, lane);
        for (int n = i + 1; n < N; ++n) {
            const int sig = spin(n);
            if constexpr (L::RIDERS) {
                // the step's accumulators carry the logits of the state that entered it (site n - 1, spin sig_in); site i is
                // not part of the sum, the last site's logits come from the VALU head behind the loop
                // (no branches in this loop body: with them hipcc moves the riders out from between the MFMAs, +19 %)
                float zp[1];
                C::step_stream(lds, C::stream_source(wsplit), sig_in, h, sf, lane, zp);
                float lq0, lq1;
                log_softmax2(zp[0], lq0, lq1);
                const float add = sig_in ? lq1 : lq0;
                lp += (double)(n > i + 1 ? add : 0.0f);
            } else {
                C::split(h, sig_in, R);
                C::step(lds, sig_in, R, h, lane);
                float z[1];
                C::head(lds, h, lane, z);
                float lp0, lp1;
                log_softmax2(z[0], lp0, lp1);
                lp += (double)(sig ? lp1 : lp0);
            }
            sig_in = sig;
        }
        if constexpr (L::RIDERS) {                            // the last site's logits: VALU head on the final state
            float z[1];
            C::head(lds, h, lane, z);
            float lp0, lp1;
            log_softmax2(z[0], lp0, lp1);
            lp += (double)(sig_in ? lp1 : lp0);
        }
        if (valid && hh == 0) {
            const int64_t row = a.row_of_pos ? a.row_of_pos[i] : i + 1;
            a.lpq[row * a.ns + s] += lp;
        }
    }
}

// Ping-pong form of the same pass (K-packed layout MODE 2, 37..50 units): 8 waves per workgroup, two per SIMD (waves w
// and w + 4 share one).  Every wave walks its own tiles (same arithmetic as above) but in lock step with the workgroup:
//     [MFMA segment of step n]  barrier  [VALU segment: gates, head, log-softmax, re-split of the new state]  barrier
// and waves 4-7 run one segment behind waves 0-3, so that on every SIMD one wave multiplies while the other does
// vector work (bf16 MFMA and VALU of DIFFERENT waves overlap on gfx950; the f32-input MFMA does not - measured,
// tools/microbench/issue_model, profiles/r02_issue_model*.txt).  Both segments are written to be issue-bound:
// split_pp.h / split_mfma_asm.h.  All waves execute the same number of barriers: the workgroup iterates to the
// largest per-wave step count; the tile walk is a snake over the length-sorted tiles, so those counts agree within a
// step or two and idle iterations (barriers only) are rare.
template <int NF32, int RJ>
__global__ void __launch_bounds__(512) prnn_flip_pp_kernel(PrnnArgs a, const void* wsplit, int kt16) {
    using PP = SplitPP<NF32, RJ, 1>;
    using C = typename PP::C;
    using L = typename C::L;
    constexpr int NU = C::NU, NT = C::NT, NB = PP::NB, WAVES = 8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ int max_steps;
    if (threadIdx.x == 0) max_steps = 0;
    C::stage(lds, wsplit);                                    // ends with __syncthreads()
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform: tile walk and branches stay scalar
    const bool late = wave >= 4;                              // the half that runs one segment behind
    const int64_t gw = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int64_t nsb32 = (a.ns + 31) / 32;
    const int64_t ntiles = (int64_t)(N - 1) * nsb32;
    const float* hck = reinterpret_cast<const float*>(a.hck);
    // tiles are sorted by chain length (flipped site ascending = longest first); round r of the walk hands wave gw the
    // tile r nw + gw in even rounds and r nw + (nw - 1 - gw) in odd ones, so every wave gets the same total length
    auto tile_of = [&](int64_t r) -> int64_t { return r * nw + ((r & 1) ? nw - 1 - gw : gw); };
    {
        int mine = 0;
        for (int64_t r = 0;; ++r) {
            const int64_t t = tile_of(r);
            if (r * nw >= ntiles) break;
            if (t < ntiles) mine += N - 1 - (int)(t / nsb32);
        }
        if (lane == 0) atomicMax(&max_steps, mine);
        __syncthreads();
    }
    const int iters = max_steps;

    // per-wave chain state
    int64_t round = 0, tile = tile_of(0);
    bool active = tile < ntiles;
    int i = 0, n = 0, sig_in = 0;
    int s = 0, sc = 0;                                        // sample index (a pass holds < 2^31 chains)
    bool valid = false;
    uint32_t word = 0, word_next = 0;
    double lp = 0.0;
    float h[NU];
    u32x4 B[NB];
    f32x16 acc[NT];
    auto load_word = [&](int w) -> uint32_t { return w * 32 < N ? a.bits[(int64_t)w * a.ns + sc] : 0u; };
    // this wave's next tile after round r (a round may hold none for it): returns false when the walk is over
    auto peek_tile = [&](int64_t r, int64_t& r_out, int64_t& t_out) -> bool {
        for (;;) {
            ++r;
            if (r * nw >= ntiles) return false;
            const int64_t t = tile_of(r);
            if (t < ntiles) { r_out = r; t_out = t; return true; }
        }
    };
    // A tile switch sits inside a VALU segment with the SIMD partner - and through the barrier the whole workgroup -
    // waiting, so it must not wait for memory.  The next tile's checkpoint (this lane's NU state values) and spin words
    // are therefore requested at the TOP of the chain's last VALU segment, into registers of their own, and consumed at
    // its end: gates + head (~3 000 cycles) cover the latency.  The values live inside one segment only - carried around
    // the loop, hipcc shuffles them through ~60 copies per iteration and waits for them at once (measured, round 2).
    float hn[NU];
    uint32_t wn0 = 0, wn1 = 0;
    auto fetch_tile = [&](int64_t tn) {
        const int in = (int)(tn / nsb32);
        const int64_t sbn = tn - (int64_t)in * nsb32;
        const int sn = (int)sbn * 32 + c;
        const int scn = sn < a.ns ? sn : (int)a.ns - 1;
        // unit u of chain sc sits at float (u >> 2) * 64 + (u & 3) * 16 of the chain's 16-chain block; the upper lane half
        // owns units shifted by a constant per group (full tiles +4, remainder +(RJ-1), special +1): three per-lane
        // base pointers and immediate offsets, instead of one 64-bit address per load (HP <= 4 kt16: host-checked)
        const float* src = hck + (((int64_t)in * a.nsb + (scn >> 4)) * kt16) * 64 + (scn & 15);
        auto off = [](int u) { return (u >> 2) * 64 + ((u & 3) << 4); };
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            const int u0 = L::unit_of(e, 0), u1 = L::unit_of(e, 1);
            const int d = off(u1) - off(u0);                   // compile-time constant per entry
            hn[e] = (src + (hh ? d : 0))[off(u0)];
        }
        const int w = in >> 5;
        wn0 = a.bits[(int64_t)w * a.ns + scn];
        wn1 = (w + 1) * 32 < N ? a.bits[(int64_t)(w + 1) * a.ns + scn] : 0u;
    };
    auto begin_tile = [&]() {                                 // tile's data are in hn / wn0 / wn1
        i = (int)(tile / nsb32);
        const int64_t sb = tile - (int64_t)i * nsb32;
        s = (int)sb * 32 + c;
        valid = s < a.ns;
        sc = valid ? s : (int)a.ns - 1;
#pragma unroll
        for (int e = 0; e < NU; ++e) h[e] = hn[e];
        word = wn0;
        word_next = wn1;
        sig_in = 1 - (int)((word >> (i & 31)) & 1);
        n = i + 1;
        if ((n & 31) == 0) { word = word_next; word_next = load_word((n >> 5) + 1); }
        lp = 0.0;
    };
    if (active) {
        fetch_tile(tile);
        begin_tile();
        PP::preload(lds, sig_in, lane, acc);
        PP::split(h, B);
    }
#ifdef RNNWF_DIAGNOSTICS      // in-kernel cycle stamps (tools/stamps.py): where a wave-step's cycles go, and the clock held
    unsigned long long t_m = 0, t_b1 = 0, t_v = 0, t_b2 = 0, t_sw = 0, t_g = 0, t_h = 0, ts = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime(), r_begin = __builtin_amdgcn_s_memrealtime();
#define RNNWF_STAMP(acc_) do { if (a.stamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - ts; ts = t_; } } while (0)
    ts = t_begin;
#else
#define RNNWF_STAMP(acc_) do { } while (0)
#endif
    if (late) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    RNNWF_STAMP(t_b2);
    for (int it = 0; it < iters; ++it) {
        if (active && !RNNWF_ABLATED(a.ablate, 1)) PP::mfma_seg(lds, B, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
        RNNWF_STAMP(t_m);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        RNNWF_STAMP(t_b1);
        if (active) {
            const int sig = (int)((word >> (n & 31)) & 1);    // before the requests below: nothing of this step waits behind them
            const bool last = n + 1 == N;
            int64_t round_nx = 0, tile_nx = 0;
            bool more = false;
            if (last) {
                more = peek_tile(round, round_nx, tile_nx);
                if (more) fetch_tile(tile_nx);
            }
            // The head rows ride in spare slots of the mixed tiles (pack_split.h): the accumulators of this step hold the
            // logits of the state that ENTERED it, i.e. of site n - 1, whose spin is sig_in.  Site i (the flipped one) is
            // not part of the sum; the chain's last site gets its logits from the VALU head below.
            if (n > i + 1 && !RNNWF_ABLATED(a.ablate, 4)) {
                float zp[1];
                PP::head_lagged(acc, zp);
                float lq0, lq1;
                log_softmax2(zp[0], lq0, lq1);
                lp += (double)(sig_in ? lq1 : lq0);
            }
            if (!RNNWF_ABLATED(a.ablate, 2)) PP::gates(lds, sig_in, acc, h, lane);
            RNNWF_STAMP(t_g);
            if (last) {
                float z[1] = {0.5f};
                if (!RNNWF_ABLATED(a.ablate, 4)) PP::head(lds, h, lane, z);
                float lp0, lp1;
                log_softmax2(z[0], lp0, lp1);
                lp += (double)(sig ? lp1 : lp0);
            }
            sig_in = sig;
            ++n;
            RNNWF_STAMP(t_h);
            if (last) {
                if (valid && hh == 0) {                        // one add per element per pass: no-return atomic, nothing to wait for
                    const int64_t row = a.row_of_pos ? a.row_of_pos[i] : i + 1;
                    unsafeAtomicAdd(&a.lpq[row * a.ns + (int64_t)s], lp);
                }
                RNNWF_STAMP(t_v);
                active = more;
                if (more) { round = round_nx; tile = tile_nx; begin_tile(); }
                RNNWF_STAMP(t_sw);
            } else if ((n & 31) == 0) {
                word = word_next;
                word_next = load_word((n >> 5) + 1);
            }
            // ONE program point defines the next step's accumulators and state quads, whichever way the chain went on
            // (two would make the register allocator copy 145 registers at the join)
            if (active) {
                PP::preload(lds, sig_in, lane, acc);           // in flight during the split
                if (!RNNWF_ABLATED(a.ablate, 8)) PP::split(h, B);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        RNNWF_STAMP(t_v);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        RNNWF_STAMP(t_b2);
    }
    if (!late) __builtin_amdgcn_s_barrier();
#ifdef RNNWF_DIAGNOSTICS
    if (a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + gw * 16;
        o[0] = t_m; o[1] = t_b1; o[2] = t_v; o[3] = t_b2; o[4] = t_sw;
        o[5] = __builtin_amdgcn_s_memtime() - t_begin; o[6] = __builtin_amdgcn_s_memrealtime() - r_begin; o[7] = (unsigned long long)iters;
        o[8] = t_g; o[9] = t_h;
    }
#endif
}

}  // namespace rnnwf
