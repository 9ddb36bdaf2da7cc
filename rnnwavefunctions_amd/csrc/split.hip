// split.hip - host side of the bf16x3 engine (split_core.h): the flip pass of the positive RNN and the swap pass of the
// complex RNN on the bf16 matrix core, incl. their ping-pong kernels.  A translation unit of its own because it is
// compiled with -fno-slp-vectorize (build.py): packed-f32 instructions stall behind bf16 MFMAs - their own wave's and the
// SIMD partner's - whereas the f32-input-MFMA kernels of prnn.hip / crnn.hip WANT the packed forms (their MFMA and VALU
// serialise anyway, so half the VALU instructions is a straight gain: config 5 0.79 -> 0.86 of the f32 pipe).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "crnn_split_kernels.h"
#include "models.h"
#include "pack.h"
#include "pack_split.h"
#include "split_kernels.h"

using namespace rnnwf;

namespace {

// ---- bf16x3 engine for the flip pass (f32 models; above 68 units: split_stream.hip) ----------------------
template <int NF32, int RJ, int WAVES, int MODE>
struct SLaunch {
    using L = SplitLayout<NF32, RJ, 1, MODE>;
    static int flip(rnnwf_handle* h, const PrnnArgs& a, int kt16) {
        const void* fn = (const void*)prnn_flip_split_kernel<NF32, RJ, WAVES, MODE>;
        if (L::HP > 4 * kt16) return h->fail(RNNWF_ERR_INVALID, "bf16x3 layout wider than the checkpoint rows (%d > %d)", L::HP, 4 * kt16);
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, L::LDS_BYTES, &bpc)) return rc;
        const int64_t ntiles = (int64_t)(a.N - 1) * ((a.ns + 31) / 32);
        const int64_t need = (ntiles + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)bpc * h->cu_count));
        TimedLaunch tl(h, 1);
        prnn_flip_split_kernel<NF32, RJ, WAVES, MODE><<<grid, WAVES * 64, L::LDS_BYTES, h->stream>>>(a, h->wsplit.p, kt16);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    // ping-pong form (8 waves per workgroup, two per SIMD, alternating MFMA / VALU segments): K-packed layouts only
    static int flip_pp(rnnwf_handle* h, const PrnnArgs& a, int kt16) {
        if constexpr (MODE == 2) {
            const void* fn = (const void*)prnn_flip_pp_kernel<NF32, RJ>;
            if (L::HP > 4 * kt16) return h->fail(RNNWF_ERR_INVALID, "bf16x3 layout wider than the checkpoint rows (%d > %d)", L::HP, 4 * kt16);
            int bpc = 0;
            if (int rc = rnnwf::blocks_per_cu(h, fn, 512, L::BYTES, &bpc)) return rc;
            const int64_t ntiles = (int64_t)(a.N - 1) * ((a.ns + 31) / 32);
            const int64_t need = (ntiles + 7) / 8;
            const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)bpc * h->cu_count));
#ifdef RNNWF_DIAGNOSTICS
            if (getenv("RNNWF_STAMPS")) {     // in-kernel cycle stamps, median over waves -> stderr (tools/stamps.py)
                PrnnArgs b = a;
                const size_t nwv = (size_t)grid * 8;
                RNNWF_HIP(h, hipMalloc((void**)&b.stamps, nwv * 128));
                RNNWF_HIP(h, hipMemsetAsync(b.stamps, 0, nwv * 128, h->stream));
                prnn_flip_pp_kernel<NF32, RJ><<<grid, 512, L::BYTES, h->stream>>>(b, h->wsplit.p, kt16, StackArgs{});
                RNNWF_HIP(h, hipStreamSynchronize(h->stream));
                std::vector<unsigned long long> st(nwv * 16);
                RNNWF_HIP(h, hipMemcpy(st.data(), b.stamps, nwv * 128, hipMemcpyDeviceToHost));
                RNNWF_HIP(h, hipFree(b.stamps));
                const char* names[10] = {"mfma_seg", "barrier_after_mfma", "valu_seg_split_part", "barrier_after_valu", "tile_switch", "total_cycles",
                                         "realtime_ticks_100MHz", "iterations", "valu_seg_gates", "valu_seg_head_logsoftmax"};
                fprintf(stderr, "RNNWF_STAMPS grid=%u waves=%zu:", grid, nwv);
                for (int k = 0; k < 10; ++k) {
                    std::vector<unsigned long long> v(nwv);
                    for (size_t w = 0; w < nwv; ++w) v[w] = st[w * 16 + k];
                    std::sort(v.begin(), v.end());
                    fprintf(stderr, " %s med %llu min %llu max %llu;", names[k], v[nwv / 2], v[0], v[nwv - 1]);
                }
                fprintf(stderr, "\n");
                return 0;
            }
#endif
            TimedLaunch tl(h, 1);
            prnn_flip_pp_kernel<NF32, RJ><<<grid, 512, L::BYTES, h->stream>>>(a, h->wsplit.p, kt16, StackArgs{});
            RNNWF_HIP(h, hipGetLastError());
            return 0;
        } else {
            return flip(h, a, kt16);
        }
    }
    static std::vector<char> pack(const rnnwf_handle* h) { return pack_split_image<NF32, RJ, 1, MODE>(h); }
    static double mfma_flops_per_step() { return (double)L::NT * L::KS * 32768.0; }   // per 32-chain wave-step
};

#define SPLIT_DISPATCH(h, EXPR)                                      \
    do {                                                             \
        switch ((h)->NFULL) {                                        \
            case 1: { using K = SLaunch<0, 10, 4, 1>; EXPR; }        \
            case 2: { using K = SLaunch<1, 2, 4, 1>; EXPR; }         \
            case 3: if ((h)->H <= 50) { using K = SLaunch<1, 9, 4, 2>; EXPR; } \
                    else { using K = SLaunch<1, 10, 4, 0>; EXPR; }   \
            case 4: { using K = SLaunch<2, 2, 4, 0>; EXPR; }         \
        }                                                            \
    } while (0)


// ---- bf16x3 engine for the swap pass (num_units <= 68) ---------------------------------------------------
template <int NF32, int RJ, int WAVES, int MODE>
struct CSLaunch {
    using L = SplitLayout<NF32, RJ, 3, MODE>;
    static int swap(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles, int kt16) {
        const void* fn = (const void*)crnn_swap_split_kernel<NF32, RJ, WAVES, MODE>;
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, L::BYTES, &bpc)) return rc;
        const int64_t need = (max_tiles + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)bpc * h->cu_count));
        TimedLaunch tl(h, 1);
        crnn_swap_split_kernel<NF32, RJ, WAVES, MODE><<<grid, WAVES * 64, L::BYTES, h->stream>>>(a, h->wsplit.p, kt16);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    // ping-pong form (8 waves per workgroup, alternating MFMA / VALU segments): K-packed layout MODE 2 only
    static int swap_pp(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles, int kt16) {
        if constexpr (MODE == 2) {
            const void* fn = (const void*)crnn_swap_pp_kernel<NF32, RJ>;
            if (L::HP > 4 * kt16) return h->fail(RNNWF_ERR_INVALID, "bf16x3 layout wider than the checkpoint rows (%d > %d)", L::HP, 4 * kt16);
            int bpc = 0;
            constexpr size_t LDS = SplitPP<NF32, RJ, 3>::LDS_WITH_SLOTS;
            if (int rc = rnnwf::blocks_per_cu(h, fn, 512, LDS, &bpc)) return rc;
            const int64_t need = (max_tiles + 7) / 8;
            const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)bpc * h->cu_count));
            TimedLaunch tl(h, 1);
            crnn_swap_pp_kernel<NF32, RJ><<<grid, 512, LDS, h->stream>>>(a, h->wsplit.p, kt16, StackArgs{});
            RNNWF_HIP(h, hipGetLastError());
            return 0;
        } else {
            return swap(h, a, max_tiles, kt16);
        }
    }
    static std::vector<char> pack(const rnnwf_handle* h) { return pack_split_image<NF32, RJ, 3, MODE>(h); }
    static double mfma_flops_per_step() { return (double)L::NT * L::KS * 32768.0; }
};

#define CSPLIT_DISPATCH(h, EXPR)                                     \
    do {                                                             \
        switch ((h)->NFULL) {                                        \
            case 1: { using K = CSLaunch<0, 10, 4, 1>; EXPR; }       \
            case 2: { using K = CSLaunch<1, 2, 4, 1>; EXPR; }        \
            case 3: if ((h)->H <= 50) { using K = CSLaunch<1, 9, 4, 2>; EXPR; } \
                    else { using K = CSLaunch<1, 10, 4, 0>; EXPR; }  \
            case 4: { using K = CSLaunch<2, 2, 4, 0>; EXPR; }        \
        }                                                            \
    } while (0)


// widths served by the riders form (split_stream.hip): 53..100 units (53..68: RNNWF_ENGINE=bf16x3-serial selects the padded
// serial kernel of round 1 instead, for A/B runs: 5.55 against 4.56 ms at N=80, 64 units, 10 000 samples)
bool riders(const rnnwf_handle* h) { return h->NFULL == 6 || (h->NFULL == 4 && h->knobs.engine != 3); }      // (engine 3: diagnostics builds only)
// 37..52 units, positive RNN: RNNWF_ENGINE=bf16x3-n16 runs the flip pass in the 16x16x32 riders form (split_stream.hip:
// prnn_flip_riders16n_asm_kernel) instead of the 32x32x16 ping-pong kernel - built and measured in round 3, 10 % slower
bool riders16n(const rnnwf_handle* h) {
#ifdef RNNWF_DIAGNOSTICS
    return h->NFULL == 3 && h->model != RNNWF_MODEL_CRNN_U1 && h->knobs.engine == 7;
#else
    return false;
#endif
}

}  // namespace

// ---- cooperative base pass on the bf16 matrix core (gru_kernels.h: coop_base_pass_bf) ---------------------------------
namespace {
template <int NFULL>
struct BfBase {
    using B = BaseBfLayout<NFULL>;
    template <int NOUT> static size_t lds_bytes() {
        return GruLayout<float, NFULL, NOUT>::BYTES + B::BYTES + (size_t)B::NB * (B::PB_BYTES + (size_t)2 * (4 * NFULL + 1) * 64 * 4 + 2 * 64 * 4);
    }
    template <typename Args, typename Kern>
    static int launch(rnnwf_handle* h, const Args& a, Kern kern, size_t lds) {
        const void* fn = (const void*)kern;
        const int threads = B::NB * (B::NW + 1) * 64;              // per block: NW product / gate waves + the sampler
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, threads, lds, &bpc)) return rc;
        // every CU gets work before any workgroup gets a second block: grid = min(blocks, CUs x resident workgroups)
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(a.nsb, (int64_t)bpc * h->cu_count));
#ifdef RNNWF_DIAGNOSTICS
        if constexpr (std::is_same<Args, PrnnArgs>::value) {
            if (getenv("RNNWF_STAMPS_BASE")) {    // in-kernel cycle stamps, median over the waves of each role -> stderr (tools/stamps_base.py)
                Args b = a;
                const size_t nwv = (size_t)grid * B::NB * (B::NW + 1);
                RNNWF_HIP(h, hipMalloc((void**)&b.stamps, nwv * 64));
                RNNWF_HIP(h, hipMemsetAsync(b.stamps, 0, nwv * 64, h->stream));
                kern<<<grid, threads, lds, h->stream>>>(b);
                RNNWF_HIP(h, hipStreamSynchronize(h->stream));
                std::vector<unsigned long long> st(nwv * 8);
                RNNWF_HIP(h, hipMemcpy(st.data(), b.stamps, nwv * 64, hipMemcpyDeviceToHost));
                RNNWF_HIP(h, hipFree(b.stamps));
                const char* names[7] = {"products", "wait_barrier_B", "gates_writes", "wait_barrier_A", "site", "total_cycles", "realtime_100MHz"};
                for (int role = 0; role < 3; ++role) {
                    fprintf(stderr, "RNNWF_STAMPS_BASE grid=%u %s waves:", grid, role == 2 ? "sampler" : role ? "remainder" : "gate");
                    for (int k = 0; k < 7; ++k) {
                        std::vector<unsigned long long> v;
                        for (size_t w = 0; w < nwv; ++w) {
                            const int mm = (int)st[w * 8 + 7], r_ = mm < NFULL ? 0 : mm == NFULL ? 1 : 2;
                            if (st[w * 8 + 5] && r_ == role) v.push_back(st[w * 8 + k]);
                        }
                        if (v.empty()) continue;
                        std::sort(v.begin(), v.end());
                        fprintf(stderr, " %s med %llu max %llu;", names[k], v[v.size() / 2], v.back());
                    }
                    fprintf(stderr, "\n");
                }
                return 0;
            }
        }
#endif
        TimedLaunch tl(h, 0);
        kern<<<grid, threads, lds, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
};
}  // namespace

bool rnnwf::base_bf_available(const rnnwf_handle* h) { return h->base_bf; }

int rnnwf::base_bf_pack(rnnwf_handle* h) {
    h->base_bf = false;
    // measured (profiles/r03_g_base_pass.md): at 37..52 units the pass gains 14 - 16 % (configs 2, 3); at 20 units and 500 samples
    // (config 1) the f32 cooperative kernel is 9 % faster - its image is three times smaller to stage - so narrower models keep it
    if (h->f64 || h->NL != 1 || h->NFULL != 3 || h->knobs.base_f32 || h->knobs.no_coop || h->knobs.engine == 1) return 0;
    std::vector<char> img;
    switch (h->NFULL) {
        case 1: img = pack_base_bf_image<1>(h); break;
        case 2: img = pack_base_bf_image<2>(h); break;
        case 3: img = pack_base_bf_image<3>(h); break;
        default: return 0;
    }
    if (int rc = ensure(h, h->wbasebf, img.size())) return rc;
    if (int rc = upload(h, h->wbasebf.p, img.data(), img.size())) return rc;
    h->base_bf = true;
    return 0;
}

int rnnwf::prnn_base_coop_bf(rnnwf_handle* h, const PrnnArgs& a0) {
    PrnnArgs a = a0;
    a.wbf = h->wbasebf.p;
    switch (h->NFULL) {
        case 1: return BfBase<1>::launch(h, a, prnn_base_coop_kernel<1, true>, BfBase<1>::lds_bytes<1>());
        case 2: return BfBase<2>::launch(h, a, prnn_base_coop_kernel<2, true>, BfBase<2>::lds_bytes<1>());
        case 3: return BfBase<3>::launch(h, a, prnn_base_coop_kernel<3, true>, BfBase<3>::lds_bytes<1>());
    }
    return h->fail(RNNWF_ERR_INVALID, "no bf16 cooperative base kernel for NFULL=%d", h->NFULL);
}

int rnnwf::crnn_base_coop_bf(rnnwf_handle* h, const CrnnArgs& a0) {
    CrnnArgs a = a0;
    a.wbf = h->wbasebf.p;
    switch (h->NFULL) {
        case 1: return BfBase<1>::launch(h, a, crnn_base_coop_kernel<1, true>, BfBase<1>::lds_bytes<3>());
        case 2: return BfBase<2>::launch(h, a, crnn_base_coop_kernel<2, true>, BfBase<2>::lds_bytes<3>());
        case 3: return BfBase<3>::launch(h, a, crnn_base_coop_kernel<3, true>, BfBase<3>::lds_bytes<3>());
    }
    return h->fail(RNNWF_ERR_INVALID, "no bf16 cooperative base kernel for NFULL=%d", h->NFULL);
}


// ---- stacked layers on the bf16x3 engine: a pipeline of one kernel per layer (split_core.h: SplitUpperLayout) -----------------
namespace {
constexpr int kStackNF32 = 1, kStackRJ = 9;                  // the K-packed layout of 37..50 units (SplitLayout MODE 2)
using StackL0 = SplitLayout<kStackNF32, kStackRJ, 1, 2>;
using StackU1 = SplitUpperLayout<kStackNF32, kStackRJ, 1>;
using StackU3 = SplitUpperLayout<kStackNF32, kStackRJ, 3>;

template <typename Kern>
int stack_grid(rnnwf_handle* h, Kern kern, size_t lds, int64_t tiles, unsigned* grid) {
    int bpc = 0;
    if (int rc = rnnwf::blocks_per_cu(h, (const void*)kern, 512, lds, &bpc)) return rc;
    *grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((tiles + 7) / 8, (int64_t)bpc * h->cu_count));
    return 0;
}
}  // namespace

bool rnnwf::stack_split_available(const rnnwf_handle* h) {
    return !h->f64 && h->NL > 1 && h->NFULL == 3 && h->H <= 50 && h->knobs.engine != 1 &&
           (h->model == RNNWF_MODEL_GRU1D || h->model == RNNWF_MODEL_GRU1D_PARITY || h->model == RNNWF_MODEL_CRNN_U1);
}
size_t rnnwf::stack_record_bytes_per_32_chains(const rnnwf_handle* h, int64_t steps) {
    return (size_t)steps * StackU1::RECORD_FLOATS * 4 * (h->NL > 2 ? 2 : 1);
}
double rnnwf::stack_split_flops_per_step(rnnwf_handle* h) {
    return ((double)StackL0::NT * StackL0::KS + (double)(h->NL - 1) * 2.0 * StackU1::NTB * StackL0::KS) * 32768.0;
}

int rnnwf::prnn_stack_pack(rnnwf_handle* h) {
    {
        const std::vector<char> img = pack_split_image<kStackNF32, kStackRJ, 1, 2>(h);
        if (int rc = ensure(h, h->wsplit, img.size())) return rc;
        if (int rc = upload(h, h->wsplit.p, img.data(), img.size())) return rc;
    }
    for (int l = 1; l < h->NL; ++l) {
        const std::vector<char> img = pack_split_upper_image<kStackNF32, kStackRJ, 1>(h, l, l == h->NL - 1);
        if (int rc = ensure(h, h->wsplit_up[l - 1], img.size())) return rc;
        if (int rc = upload(h, h->wsplit_up[l - 1].p, img.data(), img.size())) return rc;
    }
    return 0;
}

int rnnwf::prnn_stack_flip(rnnwf_handle* h, const PrnnArgs& a) {
    const int kt16 = 4 * h->NFULL + 1, NL = h->NL;
    if (StackL0::HP > 4 * kt16) return h->fail(RNNWF_ERR_INVALID, "bf16x3 layout wider than the checkpoint rows (%d > %d)", StackL0::HP, 4 * kt16);
    const int64_t nsb32 = (a.ns + 31) / 32;
    const int64_t ntiles = (int64_t)(a.N - 1) * nsb32;
    const int64_t nrec = nsb32 * (int64_t)a.N * (a.N - 1) / 2;
    const size_t bytes = (size_t)nrec * StackU1::RECORD_FLOATS * 4;
    if (int rc = ensure(h, h->xrec[0], bytes)) return rc;
    if (NL > 2) if (int rc = ensure(h, h->xrec[1], bytes)) return rc;
    unsigned g0 = 0, gu = 0, gl = 0;
    if (int rc = stack_grid(h, prnn_flip_pp_kernel<kStackNF32, kStackRJ, true>, StackL0::BYTES, ntiles, &g0)) return rc;
    if (int rc = stack_grid(h, prnn_flip_pp_upper_kernel<kStackNF32, kStackRJ, false>, StackU1::LDS_BYTES, ntiles, &gu)) return rc;
    if (int rc = stack_grid(h, prnn_flip_pp_upper_kernel<kStackNF32, kStackRJ, true>, StackU1::LDS_BYTES, ntiles, &gl)) return rc;
    TimedLaunch tl(h, 1);                                     // the whole pipeline is the "flip pass" of the timers
    StackArgs st{nullptr, (float*)h->xrec[0].p, NL * kt16, 0};
    prnn_flip_pp_kernel<kStackNF32, kStackRJ, true><<<g0, 512, StackL0::BYTES, h->stream>>>(a, h->wsplit.p, kt16, st);
    RNNWF_HIP(h, hipGetLastError());
    for (int l = 1; l < NL; ++l) {
        st.xin = (const float*)h->xrec[(l - 1) & 1].p;
        st.xout = l < NL - 1 ? (float*)h->xrec[l & 1].p : nullptr;
        st.koff = l * kt16;
#ifdef RNNWF_DIAGNOSTICS
        if (l == NL - 1 && getenv("RNNWF_STAMPS")) {          // in-kernel cycle stamps of the top layer's kernel, median over waves -> stderr (tools/stamps.py)
            PrnnArgs b = a;
            const size_t nwv = (size_t)gl * 8;
            RNNWF_HIP(h, hipMalloc((void**)&b.stamps, nwv * 128));
            RNNWF_HIP(h, hipMemsetAsync(b.stamps, 0, nwv * 128, h->stream));
            prnn_flip_pp_upper_kernel<kStackNF32, kStackRJ, true><<<gl, 512, StackU1::LDS_BYTES, h->stream>>>(b, h->wsplit_up[l - 1].p, kt16, st);
            RNNWF_HIP(h, hipStreamSynchronize(h->stream));
            std::vector<unsigned long long> sv(nwv * 16);
            RNNWF_HIP(h, hipMemcpy(sv.data(), b.stamps, nwv * 128, hipMemcpyDeviceToHost));
            RNNWF_HIP(h, hipFree(b.stamps));
            const char* names[10] = {"mfma_seg", "barrier_after_mfma", "valu_seg_split_part", "barrier_after_valu", "tile_switch", "total_cycles",
                                     "realtime_ticks_100MHz", "iterations", "valu_seg_head_gates", "valu_seg_store_head"};
            fprintf(stderr, "RNNWF_STAMPS upper kernel grid=%u waves=%zu:", gl, nwv);
            for (int k = 0; k < 10; ++k) {
                std::vector<unsigned long long> v(nwv);
                for (size_t w = 0; w < nwv; ++w) v[w] = sv[w * 16 + k];
                std::sort(v.begin(), v.end());
                fprintf(stderr, " %s med %llu min %llu max %llu;", names[k], v[nwv / 2], v[0], v[nwv - 1]);
            }
            fprintf(stderr, "\n");
            continue;
        }
#endif
        if (l < NL - 1) prnn_flip_pp_upper_kernel<kStackNF32, kStackRJ, false><<<gu, 512, StackU1::LDS_BYTES, h->stream>>>(a, h->wsplit_up[l - 1].p, kt16, st);
        else prnn_flip_pp_upper_kernel<kStackNF32, kStackRJ, true><<<gl, 512, StackU1::LDS_BYTES, h->stream>>>(a, h->wsplit_up[l - 1].p, kt16, st);
        RNNWF_HIP(h, hipGetLastError());
    }
    return 0;
}


int rnnwf::crnn_stack_pack(rnnwf_handle* h) {
    {
        const std::vector<char> img = pack_split_image<kStackNF32, kStackRJ, 3, 2>(h);
        if (int rc = ensure(h, h->wsplit, img.size())) return rc;
        if (int rc = upload(h, h->wsplit.p, img.data(), img.size())) return rc;
    }
    for (int l = 1; l < h->NL; ++l) {
        const std::vector<char> img = pack_split_upper_image<kStackNF32, kStackRJ, 3>(h, l, l == h->NL - 1);
        if (int rc = ensure(h, h->wsplit_up[l - 1], img.size())) return rc;
        if (int rc = upload(h, h->wsplit_up[l - 1].p, img.data(), img.size())) return rc;
    }
    return 0;
}

// max_records: upper bound of the wave-steps of all tiles (crnn.hip derives it from the bonds a first-changed site can have)
int rnnwf::crnn_stack_swap(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles, int64_t max_records) {
    using CL0 = SplitLayout<kStackNF32, kStackRJ, 3, 2>;
    const int kt16 = 4 * h->NFULL + 1, NL = h->NL;
    if (CL0::HP > 4 * kt16) return h->fail(RNNWF_ERR_INVALID, "bf16x3 layout wider than the checkpoint rows (%d > %d)", CL0::HP, 4 * kt16);
    const size_t bytes = (size_t)max_records * StackU3::RECORD_FLOATS * 4;
    if (int rc = ensure(h, h->xrec[0], bytes)) return rc;
    if (NL > 2) if (int rc = ensure(h, h->xrec[1], bytes)) return rc;
    unsigned g0 = 0, gu = 0, gl = 0;
    constexpr size_t LDS0 = SplitPP<kStackNF32, kStackRJ, 3>::LDS_WITH_SLOTS;
    if (int rc = stack_grid(h, crnn_swap_pp_kernel<kStackNF32, kStackRJ, true>, LDS0, max_tiles, &g0)) return rc;
    if (int rc = stack_grid(h, crnn_swap_pp_upper_kernel<kStackNF32, kStackRJ, false>, StackU3::LDS_BYTES, max_tiles, &gu)) return rc;
    if (int rc = stack_grid(h, crnn_swap_pp_upper_kernel<kStackNF32, kStackRJ, true>, StackU3::LDS_BYTES, max_tiles, &gl)) return rc;
    TimedLaunch tl(h, 1);
    StackArgs st{nullptr, (float*)h->xrec[0].p, NL * kt16, 0};
    crnn_swap_pp_kernel<kStackNF32, kStackRJ, true><<<g0, 512, LDS0, h->stream>>>(a, h->wsplit.p, kt16, st);
    RNNWF_HIP(h, hipGetLastError());
    for (int l = 1; l < NL; ++l) {
        st.xin = (const float*)h->xrec[(l - 1) & 1].p;
        st.xout = l < NL - 1 ? (float*)h->xrec[l & 1].p : nullptr;
        st.koff = l * kt16;
        if (l < NL - 1) crnn_swap_pp_upper_kernel<kStackNF32, kStackRJ, false><<<gu, 512, StackU3::LDS_BYTES, h->stream>>>(a, h->wsplit_up[l - 1].p, kt16, st);
        else crnn_swap_pp_upper_kernel<kStackNF32, kStackRJ, true><<<gl, 512, StackU3::LDS_BYTES, h->stream>>>(a, h->wsplit_up[l - 1].p, kt16, st);
        RNNWF_HIP(h, hipGetLastError());
    }
    return 0;
}

int rnnwf::prnn_split_flip(rnnwf_handle* h, const PrnnArgs& a) {
    const int kt16 = 4 * h->NFULL + 1;
    if (riders(h)) return prnn_split_flip_stream(h, a, kt16);
#ifdef RNNWF_DIAGNOSTICS
    if (riders16n(h)) return prnn_split_flip_16n(h, a, kt16);
    if (h->knobs.engine == 3) { SPLIT_DISPATCH(h, return K::flip(h, a, kt16)); }      // RNNWF_ENGINE=bf16x3-serial: A/B only
#endif
    SPLIT_DISPATCH(h, return K::flip_pp(h, a, kt16));
    return h->fail(RNNWF_ERR_INVALID, "no bf16x3 kernel for NFULL=%d", h->NFULL);
}
double rnnwf::prnn_split_flops_per_step(rnnwf_handle* h) {
    if (riders(h)) return prnn_split_stream_flops_per_step(h);
#ifdef RNNWF_DIAGNOSTICS
    if (riders16n(h)) return prnn_split_16n_flops_per_step();
#endif
    SPLIT_DISPATCH(h, return K::mfma_flops_per_step());
    return 0;
}


int rnnwf::prnn_split_pack(rnnwf_handle* h, std::vector<char>& simg) {
    if (riders(h)) return prnn_split_stream_pack(h, simg);
#ifdef RNNWF_DIAGNOSTICS
    if (riders16n(h)) {
        if (int rc = prnn_split_16n_pack(h)) return rc;
    }
#endif
    SPLIT_DISPATCH(h, { simg = K::pack(h); return 0; });
    return h->fail(RNNWF_ERR_INVALID, "no bf16x3 layout for NFULL=%d", h->NFULL);
}

int rnnwf::crnn_split_swap(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles) {
    const int kt16 = 4 * h->NFULL + 1;
    if (riders(h)) return crnn_split_swap_stream(h, a, max_tiles, kt16);
#ifdef RNNWF_DIAGNOSTICS
    if (h->knobs.engine == 3) { CSPLIT_DISPATCH(h, return K::swap(h, a, max_tiles, kt16)); }      // RNNWF_ENGINE=bf16x3-serial: A/B only
#endif
    CSPLIT_DISPATCH(h, return K::swap_pp(h, a, max_tiles, kt16));
    return h->fail(RNNWF_ERR_INVALID, "no bf16x3 cRNN kernel for NFULL=%d", h->NFULL);
}
double rnnwf::crnn_split_flops_per_step(rnnwf_handle* h) {
    if (riders(h)) return crnn_split_stream_flops_per_step(h);
    CSPLIT_DISPATCH(h, return K::mfma_flops_per_step());
    return 0;
}


int rnnwf::crnn_split_pack(rnnwf_handle* h, std::vector<char>& simg) {
    if (riders(h)) return crnn_split_stream_pack(h, simg);
    CSPLIT_DISPATCH(h, { simg = K::pack(h); return 0; });
    return h->fail(RNNWF_ERR_INVALID, "no bf16x3 layout for NFULL=%d", h->NFULL);
}
