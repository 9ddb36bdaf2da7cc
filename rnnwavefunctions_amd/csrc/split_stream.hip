// split_stream.hip - bf16x3 flip pass (positive RNN, config 5) and swap pass (complex RNN) at 69..100 units: the K-packed layout whose regular w3 fragments are read through L2
// (split_core.h: SplitLayout mode 3, SplitCore::step_stream).  A translation unit of its own: built WITHOUT
// -amdgpu-mfma-vgpr-form (build.py), so that the 160 accumulator registers live in AGPRs next to ~210 VGPRs.
#include <algorithm>

#include "models.h"
#include "pack.h"
#include "pack_split.h"
#include "crnn_split_kernels.h"
#include "split_kernels.h"

using namespace rnnwf;

namespace {
constexpr int NF32 = 3, RJ = 2, WAVES = 4;
using L = SplitLayout<NF32, RJ, 1, 3>;
using LC = SplitLayout<NF32, RJ, 3, 3>;                      // complex RNN: three head rows
static_assert(L::STREAM && L::HP == 100 && LC::STREAM, "layout mode 3 covers 100 units");
}  // namespace

int rnnwf::prnn_split_flip_stream(rnnwf_handle* h, const PrnnArgs& a, int kt16) {
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
double rnnwf::prnn_split_stream_flops_per_step() { return (double)L::NT * L::KS * 32768.0; }
int rnnwf::prnn_split_stream_pack(rnnwf_handle* h, std::vector<char>& simg) {
    simg = pack_split_image<NF32, RJ, 1, 3>(h);
    return 0;
}

// ---- swap pass of the complex RNN at 69..100 units ---------------------------------------------------------------
int rnnwf::crnn_split_swap_stream(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles, int kt16) {
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
double rnnwf::crnn_split_stream_flops_per_step() { return (double)LC::NT * LC::KS * 32768.0; }
int rnnwf::crnn_split_stream_pack(rnnwf_handle* h, std::vector<char>& simg) {
    simg = pack_split_image<NF32, RJ, 3, 3>(h);
    return 0;
}
