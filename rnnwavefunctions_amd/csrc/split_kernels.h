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
        for (int n = i + 1; n < N; ++n) {
            const int sig = spin(n);
            C::split(h, sig_in, R);
            C::step(lds, sig_in, R, h, lane);
            float z[1];
            C::head(lds, h, lane, z);
            float lp0, lp1;
            log_softmax2(z[0], lp0, lp1);
            lp += (double)(sig ? lp1 : lp0);
            sig_in = sig;
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
    auto next_tile = [&]() {                                  // advance to this wave's next tile (a round may hold none for it)
        for (;;) {
            ++round;
            if (round * nw >= ntiles) { active = false; return; }
            tile = tile_of(round);
            if (tile < ntiles) { active = true; return; }
        }
    };
    // Pulls tile tn's checkpoint lines towards L2 (2 x kt16 x 256 B per 32 chains, one 128-byte line per lane) with an
    // ORDINARY load whose value stays live in `touched` until the next begin_tile consumes it: an asm load whose
    // result register the compiler believes dead may land in a register that has been handed to something else.
    float touched = 0.0f;
    auto touch = [&](int64_t tn) {
        if (tn < ntiles) {
            const int in = (int)(tn / nsb32);
            const int64_t sbn = tn - (int64_t)in * nsb32;
            const int64_t blk = sbn * 2 + 1 < a.nsb ? sbn * 2 + 1 : a.nsb - 1;
            const float* nx = hck + (((int64_t)in * a.nsb + sbn * 2) * kt16) * 64;
            const int64_t span = ((blk - sbn * 2 + 1) * kt16) * 64;       // floats
            touched = (int64_t)lane * 32 < span ? nx[lane * 32] : 0.0f;
        }
    };
    auto begin_tile = [&]() {
        asm volatile("" :: "v"(touched));                     // the previous touch has landed (or was never issued)
        i = (int)(tile / nsb32);
        const int64_t sb = tile - (int64_t)i * nsb32;
        s = (int)sb * 32 + c;
        valid = s < a.ns;
        sc = valid ? s : (int)a.ns - 1;
        // unit u of chain sc sits at float (u >> 2) * 64 + (u & 3) * 16 of the chain's 16-chain block; the upper lane half
        // owns units shifted by a constant per group (full tiles +4, remainder +(RJ-1), special +1): three per-lane
        // base pointers and immediate offsets, instead of one 64-bit address per load (HP <= 4 kt16: host-checked)
        const float* src = hck + (((int64_t)i * a.nsb + (sc >> 4)) * kt16) * 64 + (sc & 15);
        auto off = [](int u) { return (u >> 2) * 64 + ((u & 3) << 4); };
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            const int u0 = L::unit_of(e, 0), u1 = L::unit_of(e, 1);
            const int d = off(u1) - off(u0);                   // compile-time constant per entry
            h[e] = (src + (hh ? d : 0))[off(u0)];
        }
        word = load_word(i >> 5);
        word_next = load_word((i >> 5) + 1);
        sig_in = 1 - (int)((word >> (i & 31)) & 1);
        n = i + 1;
        if ((n & 31) == 0) { word = word_next; word_next = load_word((n >> 5) + 1); }
        lp = 0.0;
        // this wave's next tile: touch its checkpoint lines now (last memory operation of the switch, so that nothing
        // here waits for it), and its begin_tile - inside a VALU segment, SIMD partner waiting at the barrier - finds
        // them in L2
        {
            int64_t r2 = round + 1;
            while (r2 * nw < ntiles && tile_of(r2) >= ntiles) ++r2;
            if (r2 * nw < ntiles) touch(tile_of(r2));
        }
    };
    if (active) {
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
            if (!RNNWF_ABLATED(a.ablate, 2)) PP::gates(lds, sig_in, acc, h, lane);
            RNNWF_STAMP(t_g);
            const int sig = (int)((word >> (n & 31)) & 1);
            float z[1] = {0.5f};
            if (!RNNWF_ABLATED(a.ablate, 4)) PP::head(lds, h, lane, z);
            float lp0, lp1;
            log_softmax2(z[0], lp0, lp1);
            lp += (double)(sig ? lp1 : lp0);
            sig_in = sig;
            ++n;
            RNNWF_STAMP(t_h);
            if (n == N) {
                if (valid && hh == 0) {
                    const int64_t row = a.row_of_pos ? a.row_of_pos[i] : i + 1;
                    a.lpq[row * a.ns + (int64_t)s] += lp;
                }
                RNNWF_STAMP(t_v);
                next_tile();
                if (active) begin_tile();
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
