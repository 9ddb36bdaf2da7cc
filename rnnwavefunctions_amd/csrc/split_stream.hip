// split_stream.hip - the "riders" form of the bf16x3 engine (split_core.h: SplitLayout mode 3, SplitCore::step_stream): one wave
// per SIMD whose VALU work rides between its own MFMAs.  Flip pass of the positive RNN and swap pass of the complex RNN at
// 69..100 units (config 5; regular w3 fragments read through L2) and at 53..68 units (whole image in LDS).  A translation unit
// of its own: built WITHOUT -amdgpu-mfma-vgpr-form (build.py), so that the accumulators live in AGPRs next to ~250 VGPRs.
#include <algorithm>

#include "models.h"
#include "pack.h"
#include "pack_split.h"
#include "crnn_split_kernels.h"
#include "split_kernels.h"

using namespace rnnwf;

namespace {
constexpr int WAVES = 4;
template <int NF32, int RJ>
struct RLaunch {
    using L = SplitLayout<NF32, RJ, 1, 3>;
    using LC = SplitLayout<NF32, RJ, 3, 3>;                  // complex RNN: three head rows
    static int flip(rnnwf_handle* h, const PrnnArgs& a, int kt16) {
#ifdef RNNWF_DIAGNOSTICS
        if constexpr (RidersStepAsm<NF32, RJ, 1, L::STREAM>::kAvailable) {
            // default: the 16x16x32 form; RNNWF_ENGINE=bf16x3-asm32: the 32x32x16 asm step; bf16x3-hipcc: the compiler-scheduled step (A/B)
            if (h->knobs.engine != 4 && h->knobs.engine != 5 && h->wsplit16.p) return flip_asm16(h, a, kt16);
            if (h->knobs.engine != 4) return flip_asm(h, a, kt16);
        }
#else
        // 69..100 units: the 16x16x32 form (generated asm step); 53..68 units: the compiler-scheduled riders step below
        if constexpr (L::STREAM) return flip_asm16(h, a, kt16);
        else
#endif
        {
        const void* fn = (const void*)prnn_flip_split_kernel<NF32, RJ, WAVES, 3>;
        if (L::HP > 4 * kt16) return h->fail(RNNWF_ERR_INVALID, "bf16x3 layout wider than the checkpoint rows (%d > %d)", L::HP, 4 * kt16);
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, L::LDS_BYTES, &bpc)) return rc;
        const int64_t ntiles = (int64_t)(a.N - 1) * ((a.ns + 31) / 32);
        const int64_t need = (ntiles + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)bpc * h->cu_count));
        TimedLaunch tl(h, 1);
        prnn_flip_split_kernel<NF32, RJ, WAVES, 3><<<grid, WAVES * 64, L::LDS_BYTES, h->stream>>>(a, h->wsplit.p, kt16);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
        }
    }
#ifdef RNNWF_DIAGNOSTICS
    // the same pass with the wave-step as one hand-scheduled asm block (split_riders_asm.h)
    static int flip_asm(rnnwf_handle* h, const PrnnArgs& a, int kt16) {
        const void* fn = (const void*)prnn_flip_riders_asm_kernel<NF32, RJ, WAVES>;
        if (L::HP > 4 * kt16) return h->fail(RNNWF_ERR_INVALID, "bf16x3 layout wider than the checkpoint rows (%d > %d)", L::HP, 4 * kt16);
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, L::LDS_BYTES, &bpc)) return rc;
        const int64_t ntiles = (int64_t)(a.N - 1) * ((a.ns + 31) / 32);
        const int64_t need = (ntiles + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)bpc * h->cu_count));
        TimedLaunch tl(h, 1);
        prnn_flip_riders_asm_kernel<NF32, RJ, WAVES><<<grid, WAVES * 64, L::LDS_BYTES, h->stream>>>(a, h->wsplit.p, kt16);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
#endif
    // the 16x16x32 form (split16_core.h), 69..100 units
    static int flip_asm16(rnnwf_handle* h, const PrnnArgs& a, int kt16) {
        using L16 = S16Layout<1>;
        const void* fn = (const void*)prnn_flip_riders16_asm_kernel<WAVES>;
        if (kt16 != L16::NJ) return h->fail(RNNWF_ERR_INVALID, "the 16x16x32 form expects %d checkpoint k-steps, got %d", L16::NJ, kt16);
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, L16::LDS_BYTES, &bpc)) return rc;
        const int64_t ntiles = (int64_t)(a.N - 1) * ((a.ns + 31) / 32);
        const int64_t need = (ntiles + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)bpc * h->cu_count));
        TimedLaunch tl(h, 1);
        prnn_flip_riders16_asm_kernel<WAVES><<<grid, WAVES * 64, L16::LDS_BYTES, h->stream>>>(a, h->wsplit16.p, kt16);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static int swap(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles, int kt16) {
        const void* fn = (const void*)crnn_swap_split_kernel<NF32, RJ, WAVES, 3>;
        if (LC::HP > 4 * kt16) return h->fail(RNNWF_ERR_INVALID, "bf16x3 layout wider than the checkpoint rows (%d > %d)", LC::HP, 4 * kt16);
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, LC::LDS_BYTES, &bpc)) return rc;
        const int64_t need = (max_tiles + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)bpc * h->cu_count));
        TimedLaunch tl(h, 1);
        crnn_swap_split_kernel<NF32, RJ, WAVES, 3><<<grid, WAVES * 64, LC::LDS_BYTES, h->stream>>>(a, h->wsplit.p, kt16);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
};
using R100 = RLaunch<3, 2>;                                  // 69..100 units (NFULL = 6 of the f32 layout)
using R68 = RLaunch<2, 2>;                                   // 53..68 units (NFULL = 4)
static_assert(R100::L::STREAM && R100::L::HP == 100 && R100::LC::STREAM, "layout mode 3 at 100 units streams its regular w3 fragments");
static_assert(!R68::L::STREAM && R68::L::HP == 68 && !R68::LC::STREAM, "layout mode 3 at 68 units is LDS-resident");
}  // namespace

#ifdef RNNWF_DIAGNOSTICS
// ---- the 16x16x32 form at 37..52 units (split16_core.h: S16nLayout): A/B only (RNNWF_ENGINE=bf16x3-n16): measured 10 % slower than the ping-pong kernel ----
int rnnwf::prnn_split_flip_16n(rnnwf_handle* h, const PrnnArgs& a, int kt16) {
    using L16 = S16nLayout<1>;
    const void* fn = (const void*)prnn_flip_riders16n_asm_kernel<WAVES>;
    if (kt16 != L16::NJ) return h->fail(RNNWF_ERR_INVALID, "the 16x16x32 form expects %d checkpoint k-steps, got %d", L16::NJ, kt16);
    int bpc = 0;
    if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, L16::BYTES, &bpc)) return rc;
    const int64_t ntiles = (int64_t)(a.N - 1) * ((a.ns + 31) / 32);
    const int64_t need = (ntiles + WAVES - 1) / WAVES;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)h->cu_count));       // one wave per SIMD
    TimedLaunch tl(h, 1);
    prnn_flip_riders16n_asm_kernel<WAVES><<<grid, WAVES * 64, L16::BYTES, h->stream>>>(a, h->wsplit16.p, kt16);
    RNNWF_HIP(h, hipGetLastError());
    return 0;
}
double rnnwf::prnn_split_16n_flops_per_step() { return (double)S16nLayout<1>::NT * S16nLayout<1>::KS * 2 * 16384.0; }
int rnnwf::prnn_split_16n_pack(rnnwf_handle* h) {
    const std::vector<char> img16 = pack_split16n_image<1>(h);
    if (int rc = ensure(h, h->wsplit16, img16.size())) return rc;
    if (int rc = upload(h, h->wsplit16.p, img16.data(), img16.size())) return rc;
    return 0;
}

#endif  // RNNWF_DIAGNOSTICS

int rnnwf::prnn_split_flip_stream(rnnwf_handle* h, const PrnnArgs& a, int kt16) {
    return h->NFULL == 6 ? R100::flip(h, a, kt16) : R68::flip(h, a, kt16);
}
double rnnwf::prnn_split_stream_flops_per_step(rnnwf_handle* h) {
    // the 16x16x32 form: 19 tiles x 19 k-steps x 2 chain sets of 16 x 16 x 32 x 2 flop per 32-chain wave-step
    if (h->NFULL == 6 && h->wsplit16.p && h->knobs.engine != 4 && h->knobs.engine != 5)      // (engines 4, 5: diagnostics builds only)
        return (double)S16Layout<1>::NT * S16Layout<1>::KS * 2 * 16384.0;
    return h->NFULL == 6 ? (double)R100::L::NT * R100::L::KS * 32768.0 : (double)R68::L::NT * R68::L::KS * 32768.0;
}
int rnnwf::prnn_split_stream_pack(rnnwf_handle* h, std::vector<char>& simg) {
    if (h->NFULL == 6) {
        simg = pack_split_image<3, 2, 1, 3>(h);
        // the 16x16x32 form's image (split16_core.h) beside it: the default flip pass at these widths
        const std::vector<char> img16 = pack_split16_image<1>(h);
        if (int rc = ensure(h, h->wsplit16, img16.size())) return rc;
        if (int rc = upload(h, h->wsplit16.p, img16.data(), img16.size())) return rc;
    } else {
        simg = pack_split_image<2, 2, 1, 3>(h);
    }
    return 0;
}

// ---- swap pass of the complex RNN -----------------------------------------------------------------------------------
int rnnwf::crnn_split_swap_stream(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles, int kt16) {
    return h->NFULL == 6 ? R100::swap(h, a, max_tiles, kt16) : R68::swap(h, a, max_tiles, kt16);
}
double rnnwf::crnn_split_stream_flops_per_step(rnnwf_handle* h) {
    return h->NFULL == 6 ? (double)R100::LC::NT * R100::LC::KS * 32768.0 : (double)R68::LC::NT * R68::LC::KS * 32768.0;
}
int rnnwf::crnn_split_stream_pack(rnnwf_handle* h, std::vector<char>& simg) {
    if (h->NFULL == 6) simg = pack_split_image<3, 2, 3, 3>(h);
    else simg = pack_split_image<2, 2, 3, 3>(h);
    return 0;
}
