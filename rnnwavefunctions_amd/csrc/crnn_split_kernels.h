// crnn_split_kernels.h - J1-J2 swap pass of the complex RNN on the bf16x3 engine (split_core.h).
#pragma once
#include "crnn_kernels.h"
#include "split_core.h"
#include "split_pp.h"
#include "split_kernels.h"

namespace rnnwf {

// J1-J2 swap pass on the bf16x3 engine: same items / outputs as crnn_swap_kernel, 32 items per wave tile
// (the tile table must have been scanned with tile_items = 32).
template <int NF32, int RJ, int WAVES, int MODE>
__global__ void __launch_bounds__(WAVES * 64, (3 * NF32 + (3 * RJ + 15) / 16) <= 5 ? 2 : 1) crnn_swap_split_kernel(CrnnArgs a, const void* wsplit, int kt16) {
    using C = SplitCore<NF32, RJ, 3, MODE>;
    using L = typename C::L;
    constexpr int NU = C::NU, NR = C::NR;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, wsplit);
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int64_t ntiles = a.tile_start[N];
    const float* hck = reinterpret_cast<const float*>(a.hck);
    for (int64_t tile = gw; tile < ntiles; tile += nw) {
        int lo = 0;
        {
            int l = 0, r = N;
            while (r - l > 1) {
                const int mid = (l + r) >> 1;
                if (a.tile_start[mid] <= tile) l = mid; else r = mid;
            }
            lo = l;
        }
        const int k = (int)(tile - a.tile_start[lo]) * 32 + c;
        const bool valid = k < a.cnt[lo];
        const SwapItem it = a.items[(int64_t)lo * a.cap + (valid ? k : 0)];
        const int64_t s = it.s;
        float h[NU];
        {
            const float* src = hck + (((int64_t)lo * a.nsb + (s >> 4)) * kt16) * 64 + (s & 15);
#pragma unroll
            for (int e = 0; e < NU; ++e) {
                const int u = hh ? L::unit_of(e, 1) : L::unit_of(e, 0);
                h[e] = u < 4 * kt16 ? src[(u >> 2) * 64 + ((u & 3) << 4)] : 0.0f;
            }
        }
        int num_up = 0;
        for (int w = 0; w < (lo >> 5); ++w) num_up += __popc(a.bits[(int64_t)w * a.ns + s]);
        uint32_t word = a.bits[(int64_t)(lo >> 5) * a.ns + s];
        num_up += __popc(word & ((1u << (lo & 31)) - 1u));
        int sig_in = 1 - (int)((word >> (lo & 31)) & 1);
        num_up += sig_in;
        double re = 0.0, im = 0.0;
        unsigned R[3][NR];
        u32x4 sf[2][L::RIDERS ? C::SFN : 1];                // (> 68 units: w3 fragments read through L2, split_core.h)
        if constexpr (L::STREAM) C::stream_first(C::stream_source(wsplit), sf, lane);
        for (int n = lo + 1; n < N; ++n) {
            if ((n & 31) == 0) word = a.bits[(int64_t)(n >> 5) * a.ns + s];
            const int sig = (int)((word >> (n & 31)) & 1) ^ (n == it.hi ? 1 : 0);
            if constexpr (L::RIDERS) {
                // the step's accumulators carry the three head rows of the state that entered it: the logits of site n - 1
                // (spin sig_in, up-spins before it num_up - sig_in); site lo is not part of the sum, the last site's logits
                // come from the VALU head
                // (no branches in this loop body: with them hipcc moves the riders out from between the MFMAs)
                float zp[3];
                C::step_stream(lds, C::stream_source(wsplit), sig_in, h, sf, lane, zp);
                float la0, la1, w0, ph0, ph1;
                crnn_site(zp, n - 1, N, num_up - sig_in, la0, la1, w0, ph0, ph1);
                const float are = sig_in ? la1 : la0, aim = sig_in ? ph1 : ph0;
                re += (double)(n > lo + 1 ? are : 0.0f);
                im += (double)(n > lo + 1 ? aim : 0.0f);
            } else {
                C::split(h, sig_in, R);
                C::step(lds, sig_in, R, h, lane);
                float z[3];
                C::head(lds, h, lane, z);
                float la0, la1, w0, ph0, ph1;
                crnn_site(z, n, N, num_up, la0, la1, w0, ph0, ph1);
                re += (double)(sig ? la1 : la0);
                im += (double)(sig ? ph1 : ph0);
            }
            num_up += sig;
            sig_in = sig;
        }
        if constexpr (L::RIDERS) {                            // the last site's logits: VALU head on the final state
            if (lo + 1 < N) {
                float z[3];
                C::head(lds, h, lane, z);
                float la0, la1, w0, ph0, ph1;
                crnn_site(z, N - 1, N, num_up - sig_in, la0, la1, w0, ph0, ph1);
                re += (double)(sig_in ? la1 : la0);
                im += (double)(sig_in ? ph1 : ph0);
            }
        }
        if (valid && hh == 0) {
            const double2 b = a.cb[(int64_t)lo * a.ns + s];
            const double2 t = a.tot[s];
            const double dre = b.x + re - t.x, dim = b.y + im - t.y;
            const double mag = exp(dre) * (double)it.coef;
            a.contrib[(int64_t)it.slot * a.ns + s] = make_double2(mag * cos(dim), mag * sin(dim));
        }
    }
}

// Ping-pong form (see prnn_flip_pp_kernel in split_kernels.h for the scheme): 8 waves per workgroup, two per SIMD, MFMA
// segment of one wave beside the VALU segment of the other, K-packed layout MODE 2 (37..50 units).  Tiles come from
// the device-side table (tile_start[lo] = first 32-item tile of first-changed site lo, longest chains first); the walk
// is the same snake, so the waves of a workgroup carry the same number of steps within a few.
// STACK: first layer of a stack (see prnn_flip_pp_kernel): interleaved checkpoint rows, a record of every step's new state for the
// layer above (record index: a.rec_start[lo] + (tile - tile_start[lo]) (N - 1 - lo) + step), no heads and no output.
template <int NF32, int RJ, bool STACK = false>
__global__ void __launch_bounds__(512) crnn_swap_pp_kernel(CrnnArgs a, const void* wsplit, int kt16, StackArgs st) {
    using PP = SplitPP<NF32, RJ, 3>;
    using C = typename PP::C;
    using L = typename C::L;
    constexpr int NU = C::NU, NT = C::NT, NB = PP::NB, WAVES = 8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ int max_steps;
    if (threadIdx.x == 0) max_steps = 0;
    C::stage(lds, wsplit);
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool late = wave >= 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int64_t ntiles = a.tile_start[N];
    const float* hck = reinterpret_cast<const float*>(a.hck);
    auto tile_of = [&](int64_t r) -> int64_t { return r * nw + ((r & 1) ? nw - 1 - gw : gw); };
    auto lo_of = [&](int64_t t) -> int {
        int l = 0, r = N;
        while (r - l > 1) {
            const int mid = (l + r) >> 1;
            if (a.tile_start[mid] <= t) l = mid; else r = mid;
        }
        return l;
    };
    {
        int mine = 0;
        for (int64_t r = 0; r * nw < ntiles; ++r) {
            const int64_t t = tile_of(r);
            if (t < ntiles) mine += N - 1 - lo_of(t);
        }
        if (lane == 0) atomicMax(&max_steps, mine);
        __syncthreads();
    }
    const int iters = max_steps;

    int64_t round = 0, tile = tile_of(0);
    bool active = tile < ntiles;
    int lo = 0, n = 0, sig_in = 0, num_up = 0, s = 0;
    bool valid = false;
    SwapItem it{};
    uint32_t word = 0;
    double re = 0.0, im = 0.0;
    float h[NU];
    u32x4 B[NB];
    f32x16 acc[NT];
    int64_t rec = 0;
    // A tile switch sits inside a VALU segment with the SIMD partner - and through the barrier the whole workgroup - waiting, and
    // a swap tile needs TWO dependent fetches (its item, then the checkpoint of the item's sample).  Round 4 (profiles/
    // r04_f_sq_cfg3_summary.txt: 28 % of the wave cycles waiting on memory, 218 wave-steps per SIMD with 48 switches per
    // workgroup): the NEXT tile's item is fetched a whole tile ahead (at the switch into the current one), its checkpoint travels
    // by LDS-DMA (no register destination) into a per-wave staging slot during the current tile's last VALU segment, where the
    // chain's swap base and total are requested too; the switch itself then only reads LDS.
    constexpr size_t SLOT_BYTES = (size_t)NU * 256;
    char* slot = lds + ((L::BYTES + 15) / 16) * 16 + (size_t)wave * SLOT_BYTES;
    typedef __attribute__((address_space(3))) void* LdsVoid;
    typedef const __attribute__((address_space(1))) void* GlobVoid;
    int64_t tile_nx = -1, round_nx = 0;
    int lo_nx = 0;
    bool valid_nx = false;
    SwapItem it_nx{};
    auto fetch_item = [&](int64_t t, int& lo_t, bool& valid_t, SwapItem& it_t) {
        lo_t = lo_of(t);
        const int k = (int)(t - a.tile_start[lo_t]) * 32 + c;
        valid_t = k < a.cnt[lo_t];
        it_t = a.items[(int64_t)lo_t * a.cap + (valid_t ? k : 0)];
    };
    auto peek = [&]() {                                        // the tile after the current one, and its item
        tile_nx = -1;
        for (int64_t r = round + 1; r * nw < ntiles; ++r) {
            const int64_t t = tile_of(r);
            if (t < ntiles) { tile_nx = t; round_nx = r; break; }
        }
        if (tile_nx >= 0) fetch_item(tile_nx, lo_nx, valid_nx, it_nx);
    };
    auto dma_checkpoint = [&](int lo_t, int s_t) {             // entry e of every lane -> slot + 256 e + 4 lane
        const float* src = hck + (((int64_t)lo_t * a.nsb + (s_t >> 4)) * (STACK ? st.kstride : kt16) + (STACK ? st.koff : 0)) * 64 + (s_t & 15);
        auto off = [](int u) { return (u >> 2) * 64 + ((u & 3) << 4); };
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            const int u0 = L::unit_of(e, 0), u1 = L::unit_of(e, 1);
            const int d = off(u1) - off(u0);                   // HP <= 4 kt16: host-checked
            __builtin_amdgcn_global_load_lds((GlobVoid)(src + (hh ? d : 0) + off(u0)), (LdsVoid)(slot + e * 256), 4, 0, 0);
        }
    };
    auto wait_vm = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); };
    // tile, lo, valid, it are set and the tile's checkpoint has landed in the slot
    auto enter_tile = [&]() {
        s = it.s;
        const float* p = reinterpret_cast<const float*>(slot) + lane;
#pragma unroll
        for (int e = 0; e < NU; ++e) h[e] = p[e * 64];
        if constexpr (STACK) rec = a.rec_start[lo] + (tile - a.tile_start[lo]) * (int64_t)(N - 1 - lo);
        num_up = 0;
        for (int w = 0; w < (lo >> 5); ++w) num_up += __popc(a.bits[(int64_t)w * a.ns + s]);
        word = a.bits[(int64_t)(lo >> 5) * a.ns + s];
        num_up += __popc(word & ((1u << (lo & 31)) - 1u));
        sig_in = 1 - (int)((word >> (lo & 31)) & 1);
        num_up += sig_in;
        n = lo + 1;
        if ((n & 31) == 0 && n < N) word = a.bits[(int64_t)(n >> 5) * a.ns + s];
        re = 0.0; im = 0.0;
    };
    if (active) {
        fetch_item(tile, lo, valid, it);
        dma_checkpoint(lo, it.s);
        wait_vm();
        enter_tile();
        peek();
        PP::preload(lds, sig_in, lane, acc);
        PP::split(h, B);
    }
    if (late) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    for (int itn = 0; itn < iters; ++itn) {
        if (active) PP::mfma_seg(lds, B, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
            const bool last = n + 1 == N;
            double2 b_pre = make_double2(0.0, 0.0), t_pre = make_double2(0.0, 0.0);
            if (last) {                                        // requested now, used at the end of the segment
                if (!STACK && valid && hh == 0) {
                    b_pre = a.cb[(int64_t)lo * a.ns + s];
                    t_pre = a.tot[s];
                }
                if (tile_nx >= 0) dma_checkpoint(lo_nx, it_nx.s);
            }
            asm volatile("" ::: "memory");
            // The three head rows ride in spare slots of the mixed tiles (pack_split.h): this step's accumulators hold the
            // logits of the state that ENTERED it, i.e. of site n - 1 (spin sig_in, up-spins before it num_up - sig_in).
            // Site lo is not part of the sum; the chain's last site gets its logits from the VALU head below.
            if (!STACK && n > lo + 1) {
                float zp[3];
                PP::head_lagged(acc, zp);
                float la0, la1, w0, ph0, ph1;
                crnn_site(zp, n - 1, N, num_up - sig_in, la0, la1, w0, ph0, ph1);
                re += (double)(sig_in ? la1 : la0);
                im += (double)(sig_in ? ph1 : ph0);
            }
            PP::gates(lds, sig_in, acc, h, lane);
            if constexpr (STACK) {
                store_record<NU>(st.xout, rec, h, lane);
                ++rec;
            }
            const int sig = (int)((word >> (n & 31)) & 1) ^ (n == it.hi ? 1 : 0);
            if (!STACK && last) {
                float z[3];
                PP::head(lds, h, lane, z);
                float la0, la1, w0, ph0, ph1;
                crnn_site(z, n, N, num_up, la0, la1, w0, ph0, ph1);
                re += (double)(sig ? la1 : la0);
                im += (double)(sig ? ph1 : ph0);
            }
            num_up += sig;
            sig_in = sig;
            ++n;
            if (last) {
                if (!STACK && valid && hh == 0) {
                    const double dre = b_pre.x + re - t_pre.x, dim = b_pre.y + im - t_pre.y;
                    const double mag = exp(dre) * (double)it.coef;
                    a.contrib[(int64_t)it.slot * a.ns + s] = make_double2(mag * cos(dim), mag * sin(dim));
                }
                active = tile_nx >= 0;
                if (active) {
                    round = round_nx; tile = tile_nx; lo = lo_nx; valid = valid_nx; it = it_nx;
                    wait_vm();                                 // the checkpoint has landed
                    enter_tile();
                    peek();                                    // the item of the tile after this one: used a whole tile later
                }
            } else if ((n & 31) == 0) {
                word = a.bits[(int64_t)(n >> 5) * a.ns + s];
            }
            if (active) {
                PP::preload(lds, sig_in, lane, acc);
                PP::split(h, B);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!late) __builtin_amdgcn_s_barrier();
}

// One GRU layer above the first of the complex RNN's stack (its default: two layers, J1J2/ComplexRNNwavefunction.py:16,40), ping-pong
// form: prnn_flip_pp_upper_kernel's step (split_kernels.h) on the swap pass's tiles.  LAST: three head rows (amplitude logit
// difference, two phase logits), the U(1) mask's running up-spin count, and the item's contribution H exp(log psi(s') - log psi(s)).
template <int NF32, int RJ, bool LAST>
__global__ void __launch_bounds__(512) crnn_swap_pp_upper_kernel(CrnnArgs a, const void* wup, int kt16, StackArgs st) {
    using PU = SplitPPUpper<NF32, RJ, 3>;
    using U = typename PU::U;
    using L = typename PU::L;
    constexpr int NU = PU::NU, NTA = PU::NTA, NB = PU::NB, WAVES = 8, REC = U::RECORD_FLOATS, NG = U::NG;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ int max_steps;
    if (threadIdx.x == 0) max_steps = 0;
    PU::stage(lds, wup);
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool late = wave >= 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int64_t ntiles = a.tile_start[N];
    const float* hck = reinterpret_cast<const float*>(a.hck);
    char* slot = lds + ((U::BYTES + 15) / 16) * 16 + (size_t)wave * U::SLOT_BYTES;
    auto tile_of = [&](int64_t r) -> int64_t { return r * nw + ((r & 1) ? nw - 1 - gw : gw); };
    auto lo_of = [&](int64_t t) -> int {
        int l = 0, r = N;
        while (r - l > 1) {
            const int mid = (l + r) >> 1;
            if (a.tile_start[mid] <= t) l = mid; else r = mid;
        }
        return l;
    };
    {
        int mine = 0;
        for (int64_t r = 0; r * nw < ntiles; ++r) {
            const int64_t t = tile_of(r);
            if (t < ntiles) mine += N - 1 - lo_of(t);
        }
        if (lane == 0) atomicMax(&max_steps, mine);
        __syncthreads();
    }
    const int iters = max_steps;

    int64_t round = 0, tile = tile_of(0), rec = 0;
    bool active = tile < ntiles;
    int lo = 0, n = 0, sig_in = 0, num_up = 0, s = 0;
    bool valid = false;
    SwapItem it{};
    uint32_t word = 0;
    double re = 0.0, im = 0.0;
    float h[NU];
    u32x4 BX[NB], BH[NB];
    f32x16 acc[NTA];
    typedef __attribute__((address_space(3))) void* LdsVoid;
    typedef const __attribute__((address_space(1))) void* GlobVoid;
    auto dma_record = [&](int64_t r) {
        const float* base = st.xin + r * (int64_t)REC;
#pragma unroll
        for (int g = 0; g < NG; ++g) __builtin_amdgcn_global_load_lds((GlobVoid)(base + g * 256 + lane * 4), (LdsVoid)(slot + g * 1024), 16, 0, RNNWF_RECORD_AUX);
#pragma unroll
        for (int t = 0; t < U::NTAIL; ++t)
            __builtin_amdgcn_global_load_lds((GlobVoid)(base + NG * 256 + t * 64 + lane), (LdsVoid)(slot + NG * 1024 + t * 256), 4, 0, 0);
    };
    auto read_record = [&](float (&x)[NU]) {
        const float4* p = reinterpret_cast<const float4*>(slot) + lane;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float4 v = p[g * 64];
            x[4 * g] = v.x; x[4 * g + 1] = v.y; x[4 * g + 2] = v.z; x[4 * g + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < U::NTAIL; ++t) x[4 * NG + t] = reinterpret_cast<const float*>(slot + NG * 1024)[t * 64 + lane];
    };
    auto wait_vm = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); };
    auto wait_lds = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); asm volatile("" ::: "memory"); };
    // Tile switch as in crnn_swap_pp_kernel: the next tile's item is fetched a tile ahead, its checkpoint travels by LDS-DMA into the
    // staging slot during the current tile's last VALU segment (the slot is free there: a chain's last step requests no record);
    // at the switch the checkpoint is read out of LDS and the new tile's first record is requested - the one latency that shows.
    int64_t tile_nx = -1, round_nx = 0;
    int lo_nx = 0;
    bool valid_nx = false;
    SwapItem it_nx{};
    auto fetch_item = [&](int64_t t, int& lo_t, bool& valid_t, SwapItem& it_t) {
        lo_t = lo_of(t);
        const int k = (int)(t - a.tile_start[lo_t]) * 32 + c;
        valid_t = k < a.cnt[lo_t];
        it_t = a.items[(int64_t)lo_t * a.cap + (valid_t ? k : 0)];
    };
    auto peek = [&]() {
        tile_nx = -1;
        for (int64_t r = round + 1; r * nw < ntiles; ++r) {
            const int64_t t = tile_of(r);
            if (t < ntiles) { tile_nx = t; round_nx = r; break; }
        }
        if (tile_nx >= 0) fetch_item(tile_nx, lo_nx, valid_nx, it_nx);
    };
    auto dma_checkpoint = [&](int lo_t, int s_t) {             // entry e of every lane -> slot + 256 e + 4 lane
        const float* src = hck + (((int64_t)lo_t * a.nsb + (s_t >> 4)) * st.kstride + st.koff) * 64 + (s_t & 15);
        auto off = [](int u) { return (u >> 2) * 64 + ((u & 3) << 4); };
#pragma unroll
        for (int e = 0; e < NU; ++e) {
            const int u0 = L::unit_of(e, 0), u1 = L::unit_of(e, 1);
            const int d = off(u1) - off(u0);
            __builtin_amdgcn_global_load_lds((GlobVoid)(src + (hh ? d : 0) + off(u0)), (LdsVoid)(slot + e * 256), 4, 0, 0);
        }
    };
    // tile, lo, valid, it are set and the tile's checkpoint has landed in the slot
    auto begin_tile = [&]() {
        s = it.s;
        {
            const float* p = reinterpret_cast<const float*>(slot) + lane;
#pragma unroll
            for (int e = 0; e < NU; ++e) h[e] = p[e * 64];
        }
        wait_lds();
        rec = a.rec_start[lo] + (tile - a.tile_start[lo]) * (int64_t)(N - 1 - lo);
        dma_record(rec);
        n = lo + 1;
        if constexpr (LAST) {
            num_up = 0;
            for (int w = 0; w < (lo >> 5); ++w) num_up += __popc(a.bits[(int64_t)w * a.ns + s]);
            word = a.bits[(int64_t)(lo >> 5) * a.ns + s];
            num_up += __popc(word & ((1u << (lo & 31)) - 1u));
            sig_in = 1 - (int)((word >> (lo & 31)) & 1);
            num_up += sig_in;
            if ((n & 31) == 0 && n < N) word = a.bits[(int64_t)(n >> 5) * a.ns + s];
            re = 0.0; im = 0.0;
        }
    };
    auto load_quads = [&](bool fresh) {                       // BH <- h, BX <- the record in flight
        PU::split(h, BH);
        // behind the record's transfer only this segment's NG record stores may still be on their way (a fresh tile: wait for all)
        if (LAST || fresh) __builtin_amdgcn_s_waitcnt(0x0F70);
        else __builtin_amdgcn_s_waitcnt(0x0F70 | (NG + U::NTAIL));
        asm volatile("" ::: "memory");
        read_record(h);
        PU::split(h, BX);
        wait_lds();
    };
    if (active) {
        fetch_item(tile, lo, valid, it);
        dma_checkpoint(lo, it.s);
        wait_vm();
        begin_tile();
        peek();
        load_quads(true);
    }
    if (late) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    for (int itn = 0; itn < iters; ++itn) {
        if (active) PU::mfma_seg(lds, BX, BH, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
            const bool last = n + 1 == N;
            double2 b_pre = make_double2(0.0, 0.0), t_pre = make_double2(0.0, 0.0);
            if (!last) {
                dma_record(rec + 1);
            } else {
                if (LAST && valid && hh == 0) {                // the chain's swap base and total: requested now, used at the segment's end
                    b_pre = a.cb[(int64_t)lo * a.ns + s];
                    t_pre = a.tot[s];
                }
                if (tile_nx >= 0) dma_checkpoint(lo_nx, it_nx.s);
            }
            asm volatile("" ::: "memory");
            if constexpr (LAST) {
                // head rows of the state that ENTERED the step: site n - 1 (spin sig_in, up-spins before it num_up - sig_in)
                if (n > lo + 1) {
                    float zp[3];
                    PU::head_lagged(acc, zp);
                    float la0, la1, w0, ph0, ph1;
                    crnn_site(zp, n - 1, N, num_up - sig_in, la0, la1, w0, ph0, ph1);
                    re += (double)(sig_in ? la1 : la0);
                    im += (double)(sig_in ? ph1 : ph0);
                }
            }
            PU::gates_rem(acc, BH, h);
            PU::gates_full(acc, BH, h);
            if constexpr (!LAST) store_record<NU>(st.xout, rec, h, lane);
            if constexpr (LAST) {
                const int sig = (int)((word >> (n & 31)) & 1) ^ (n == it.hi ? 1 : 0);
                if (last) {
                    float z[3];
                    PU::head(lds, h, lane, z);
                    float la0, la1, w0, ph0, ph1;
                    crnn_site(z, n, N, num_up, la0, la1, w0, ph0, ph1);
                    re += (double)(sig ? la1 : la0);
                    im += (double)(sig ? ph1 : ph0);
                    if (valid && hh == 0) {
                        const double dre = b_pre.x + re - t_pre.x, dim = b_pre.y + im - t_pre.y;
                        const double mag = exp(dre) * (double)it.coef;
                        a.contrib[(int64_t)it.slot * a.ns + s] = make_double2(mag * cos(dim), mag * sin(dim));
                    }
                }
                num_up += sig;
                sig_in = sig;
            }
            ++n;
            ++rec;
            if (last) {
                active = tile_nx >= 0;
                if (active) {
                    round = round_nx; tile = tile_nx; lo = lo_nx; valid = valid_nx; it = it_nx;
                    wait_vm();                                 // the checkpoint has landed (and this chain's stores have gone)
                    begin_tile();
                    peek();
                }
            } else if constexpr (LAST) {
                if ((n & 31) == 0) word = a.bits[(int64_t)(n >> 5) * a.ns + s];
            }
            if (active) load_quads(last);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!late) __builtin_amdgcn_s_barrier();
}

}  // namespace rnnwf
