// crnn.hip - host side of the complex GRU RNN wave function with the U(1) mask (model CRNN_U1):
// sample / log_amplitude / fused J1-J2 local energies / fused VMC step.
#include <algorithm>

#include <cstdlib>

#include "crnn_kernels.h"
#include "crnn_ml_kernels.h"
#include "models.h"
#include "pack.h"

using namespace rnnwf;

namespace {

constexpr size_t kHckBudget = (size_t)48 << 30;
constexpr int64_t kChunk = (int64_t)1 << 20;

template <int NFULL, int WAVES>
struct CLaunch {
    using L = GruLayout<float, NFULL, 3>;
    static int blocks_per_cu(rnnwf_handle* h, const void* fn, int* out) { return rnnwf::blocks_per_cu(h, fn, WAVES * 64, L::LDS_BYTES, out); }
    static int base_coop(rnnwf_handle* h, const CrnnArgs& a) {
        if constexpr (NFULL <= 4) {
            const void* fn = (const void*)crnn_base_coop_kernel<NFULL>;
            const size_t lds = L::BYTES + (size_t)2 * L::KT * 64 * 4 + 2 * 64 * 4;
            int bpc = 0;
            if (int rc = rnnwf::blocks_per_cu(h, fn, (NFULL + 1) * 64, lds, &bpc)) return rc;
            const unsigned grid = (unsigned)std::min<int64_t>(a.nsb, (int64_t)bpc * h->cu_count);
            TimedLaunch tl(h, 0);
            crnn_base_coop_kernel<NFULL><<<grid, (NFULL + 1) * 64, lds, h->stream>>>(a);
            RNNWF_HIP(h, hipGetLastError());
        }
        return 0;
    }
    static int base(rnnwf_handle* h, const CrnnArgs& a) {
        if (NFULL <= 3 && base_bf_available(h)) return crnn_base_coop_bf(h, a);      // bf16 cooperative kernel, every batch size (prnn.hip)
        // fewer 16-chain blocks than SIMDs: the cooperative kernel (NFULL + 1 waves per block, bit-identical)
        if (NFULL <= 4 && a.nsb <= (int64_t)4 * h->cu_count && !h->knobs.no_coop) return base_coop(h, a);
        const void* fn = (const void*)crnn_base_kernel<NFULL, WAVES>;
        int bpc = 0;
        if (int rc = blocks_per_cu(h, fn, &bpc)) return rc;
        const int64_t need = (a.nsb + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
        TimedLaunch tl(h, 0);
        crnn_base_kernel<NFULL, WAVES><<<grid, WAVES * 64, L::LDS_BYTES, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static int swap(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles) {
        const void* fn = (const void*)crnn_swap_kernel<NFULL, WAVES>;
        int bpc = 0;
        if (int rc = blocks_per_cu(h, fn, &bpc)) return rc;
        // the tile count lives on the device: launch the persistent grid, bounded by the worst case
        const int64_t need = (max_tiles + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)bpc * h->cu_count));
        TimedLaunch tl(h, 1);
        crnn_swap_kernel<NFULL, WAVES><<<grid, WAVES * 64, L::LDS_BYTES, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static std::vector<char> pack(const rnnwf_handle* h) { return pack_gru_image<float, NFULL, 3>(h); }
    static size_t hck_bytes_per_block() { return (size_t)L::KT * 64 * sizeof(float); }
};

// stacked layers (forward passes on the f32-input MFMA; crnn_ml_kernels.h)
template <int NFULL, int NL, int WAVES>
struct CMLaunch {
    using M = CrnnMlCore<NFULL, NL>;
    static int base(rnnwf_handle* h, const CrnnArgs& a) {
        // all layers' images resident in LDS (up to 52 units): the gate tiles of every layer spread over NFULL + 1 waves per block of
        // 16 chains (ml_coop.h) - for every batch size, its accumulation order is not the one-wave kernel's; RNNWF_NO_COOP=1 keeps that one
        if constexpr (MlCoopLayout<NFULL, NL, 3>::FITS && M::SPILL == 0) {
            if (!h->knobs.no_coop) {
                using ML = MlCoopLayout<NFULL, NL, 3>;
                const void* cfn = (const void*)crnn_base_coop_kernel<NFULL, false, NL>;
                int cb = 0;
                if (int rc = rnnwf::blocks_per_cu(h, cfn, ML::THREADS, ML::LDS, &cb)) return rc;
                const int64_t need = (a.nsb + ML::NB - 1) / ML::NB;
                const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)cb * h->cu_count));
                TimedLaunch tl(h, 0);
                crnn_base_coop_kernel<NFULL, false, NL><<<grid, ML::THREADS, ML::LDS, h->stream>>>(a);
                RNNWF_HIP(h, hipGetLastError());
                return 0;
            }
        }
        const void* fn = (const void*)crnn_ml_base_kernel<NFULL, NL, WAVES>;
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, M::BYTES, &bpc)) return rc;
        // as few waves per workgroup as still cover the batch with every resident workgroup busy (prnn.hip: MLaunchL::base)
        const int64_t slots = (int64_t)bpc * h->cu_count;
        const int wpb = (int)std::max<int64_t>(1, std::min<int64_t>(WAVES, (a.nsb + slots - 1) / slots));
        const int64_t need = (a.nsb + wpb - 1) / wpb;
        const unsigned grid = (unsigned)std::min<int64_t>(need, slots);
        TimedLaunch tl(h, 0);
        crnn_ml_base_kernel<NFULL, NL, WAVES><<<grid, wpb * 64, M::BYTES, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static int swap(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles) {
        const void* fn = (const void*)crnn_ml_swap_kernel<NFULL, NL, WAVES>;
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, M::BYTES, &bpc)) return rc;
        const int64_t need = (max_tiles + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)bpc * h->cu_count));
        TimedLaunch tl(h, 1);
        crnn_ml_swap_kernel<NFULL, NL, WAVES><<<grid, WAVES * 64, M::BYTES, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static std::vector<char> pack(const rnnwf_handle* h) {
        std::vector<char> img = pack_gru_image<float, NFULL, 3>(h);
        for (int l = 1; l < NL; ++l) {
            const std::vector<char> up = pack_upper_image<NFULL>(h, l);
            img.insert(img.end(), up.begin(), up.end());
        }
        return img;
    }
    static size_t hck_bytes_per_block() { return (size_t)NL * M::KT * 64 * sizeof(float); }
};

#define CRNN_DISPATCH(h, EXPR)                                  \
    do {                                                        \
        if ((h)->NL == 2) {                                     \
            switch ((h)->NFULL) {                               \
                case 1: { using K = CMLaunch<1, 2, 4>; EXPR; }  \
                case 2: { using K = CMLaunch<2, 2, 4>; EXPR; }  \
                case 3: { using K = CMLaunch<3, 2, 8>; EXPR; }  \
                case 4: { using K = CMLaunch<4, 2, 4>; EXPR; }  \
                case 6: { using K = CMLaunch<6, 2, 4>; EXPR; }  \
            }                                                   \
            break;                                              \
        }                                                       \
        if ((h)->NL == 4) {                                     \
            switch ((h)->NFULL) {                               \
                case 1: { using K = CMLaunch<1, 4, 4>; EXPR; }  \
                case 2: { using K = CMLaunch<2, 4, 4>; EXPR; }  \
                case 3: { using K = CMLaunch<3, 4, 4>; EXPR; }  \
                case 4: { using K = CMLaunch<4, 4, 4>; EXPR; }  \
                case 6: { using K = CMLaunch<6, 4, 4>; EXPR; }  \
            }                                                   \
            break;                                              \
        }                                                       \
        if ((h)->NL == 3) {                                     \
            switch ((h)->NFULL) {                               \
                case 1: { using K = CMLaunch<1, 3, 4>; EXPR; }  \
                case 2: { using K = CMLaunch<2, 3, 8>; EXPR; }  \
                case 3: { using K = CMLaunch<3, 3, 8>; EXPR; }  \
                case 4: { using K = CMLaunch<4, 3, 4>; EXPR; }  \
                case 6: { using K = CMLaunch<6, 3, 4>; EXPR; }  \
            }                                                   \
            break;                                              \
        }                                                       \
        switch ((h)->NFULL) {                                   \
            case 1: { using K = CLaunch<1, 4>; EXPR; }          \
            case 2: { using K = CLaunch<2, 4>; EXPR; }          \
            case 3: { using K = CLaunch<3, 4>; EXPR; }          \
            case 4: { using K = CLaunch<4, 4>; EXPR; }          \
            case 6: { using K = CLaunch<6, 8>; EXPR; }         \
            case 8: { using K = CLaunch<8, 4>; EXPR; }          \
            case 12: { using K = CLaunch<12, 4>; EXPR; }        \
            case 16: { using K = CLaunch<16, 4>; EXPR; }        \
        }                                                       \
    } while (0)

int launch_base(rnnwf_handle* h, const CrnnArgs& a) {
    CRNN_DISPATCH(h, return K::base(h, a));
    return h->fail(RNNWF_ERR_INVALID, "no cRNN kernel for NFULL=%d", h->NFULL);
}
int launch_swap(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles) {
    CRNN_DISPATCH(h, return K::swap(h, a, max_tiles));
    return h->fail(RNNWF_ERR_INVALID, "no cRNN kernel for NFULL=%d", h->NFULL);
}
size_t hck_bytes_per_block(rnnwf_handle* h) {
    CRNN_DISPATCH(h, return K::hck_bytes_per_block());
    return 0;
}

// sites with a stored state: a stack keeps all N (the layer-wise gradient reads the lower layers' last site too)
int hck_sites(const rnnwf_handle* h) { return h->NL > 1 ? h->N : std::max(h->N - 1, 1); }

CrnnArgs base_args(rnnwf_handle* h, int64_t ns) {
    CrnnArgs a{};
    a.wimg = h->wimg.p;
    a.N = h->N;
    a.ns = ns;
    a.nsb = (ns + kChains - 1) / kChains;
    return a;
}

int64_t max_chains_per_pass(rnnwf_handle* h) {
    size_t per_block = (size_t)hck_sites(h) * hck_bytes_per_block(h);
    if (h->NL > 1 && h->engine_split)                         // the layer pipeline's records: ~2 items per sample and site, per 16 chains
        per_block += stack_record_bytes_per_32_chains(h, (int64_t)h->N * (h->N - 1) / 2) + stack_record_bytes_per_32_chains(h, 4 * (int64_t)h->N);
    return std::max<int64_t>(1, (int64_t)(state_budget_bytes(h, kHckBudget) / per_block)) * kChains;
}

// Upper bound of the wave-steps (32-item tiles x chain length) of one swap pass: first-changed site lo owns the bonds (lo, lo + 1) and
// (lo, lo + 2), with periodic couplings also (N - 1, 0), (N - 2, 0) at lo = 0 and (N - 1, 1) at lo = 1 (j1j2_enumerate_kernel) - at most
// 4, 3, 2, 2, ... items per sample.
int64_t stack_max_records(int N, int64_t ns) {
    int64_t r = 0;
    for (int lo = 0; lo < N - 1; ++lo) r += ((int64_t)(lo == 0 ? 4 : lo == 1 ? 3 : 2) * ns + 31) / 32 * (N - 1 - lo);
    return std::max<int64_t>(r, 1);
}

// J1-J2 local energies of the ns chains whose packed spins are in h->bits (drawn here when `sampling`).
// couplings_dev: J1 (N), J2 (N), Bz (N).  Leaves complex64 E_loc in h->eloc; *ncon_host (optional) receives
// the number of scored configurations after a stream sync.
int j1j2_on_device(rnnwf_handle* h, int64_t ns, bool sampling, uint64_t seed, uint64_t step, int64_t offset,
                   const double* couplings_dev, int periodic, int marshall) {
    const int N = h->N;
    const int64_t nsb = (ns + kChains - 1) / kChains;
    const int64_t cap = 4 * ns;
    if (int rc = ensure(h, h->hck, (size_t)hck_sites(h) * nsb * hck_bytes_per_block(h))) return rc;
    if (int rc = ensure(h, h->cbase, (size_t)N * ns * sizeof(double2))) return rc;
    if (int rc = ensure(h, h->cout, (size_t)ns * (sizeof(double2) + sizeof(double)))) return rc;
    if (int rc = ensure(h, h->tile_count, (size_t)(2 * N + 8) * 4 + 64 + (size_t)N * 8)) return rc;
    if (int rc = ensure(h, h->tiles, (size_t)N * cap * sizeof(SwapItem))) return rc;
    if (int rc = ensure(h, h->lpq, (size_t)ns * 2 * N * sizeof(double2))) return rc;
    if (int rc = ensure(h, h->eloc, (size_t)ns * sizeof(float2))) return rc;

    double2* tot = (double2*)h->cout.p;
    double* diag = (double*)((char*)h->cout.p + (size_t)ns * sizeof(double2));
    int32_t* cnt = (int32_t*)h->tile_count.p;
    int32_t* tile_start = cnt + N;
    int64_t* total_items = (int64_t*)((char*)h->tile_count.p + (((size_t)(2 * N + 1) * 4 + 15) / 16) * 16);
    const bool stack = h->engine_split && h->NL > 1;
    int64_t* rec_start = stack ? total_items + 4 : nullptr;     // [N] first record of each site's tiles (layer pipeline)

    CrnnArgs a = base_args(h, ns);
    a.bits = (uint32_t*)h->bits.p;
    a.hck = h->hck.p;
    a.cb = (double2*)h->cbase.p;
    a.tot = tot;
    a.sampling = sampling ? 1 : 0;
    a.seed = seed; a.step = step; a.sample_offset = offset;
    if (int rc = launch_base(h, a)) return rc;

    // the item counters are zeroed by the previous step's assembly kernel (j1j2_eloc_kernel); a memset only when that did not happen
    if (!h->j1j2_cnt_clean) RNNWF_HIP(h, hipMemsetAsync(cnt, 0, (size_t)N * 4, h->stream));
    h->j1j2_cnt_clean = false;
    J1J2Args e{};
    e.bits = (const uint32_t*)h->bits.p;
    e.ns = ns; e.N = N;
    e.J1 = couplings_dev; e.J2 = couplings_dev + N; e.Bz = couplings_dev + 2 * N;
    e.periodic = periodic; e.marshall = marshall;
    e.cnt = cnt; e.items = (SwapItem*)h->tiles.p; e.cap = cap;
    e.contrib = (double2*)h->lpq.p; e.diag = diag;
    {
        TimedLaunch tl(h, 2);
        dim3 grid((unsigned)((ns + 255) / 256), (unsigned)(2 * N));
        j1j2_enumerate_kernel<<<grid, 256, 0, h->stream>>>(e);
        RNNWF_HIP(h, hipGetLastError());
        // the three totals land in pinned[64..88) - written by the kernel itself - and are read at the caller's next stream sync
        // (collect_totals)
        j1j2_tile_scan_kernel<<<1, 64, 0, h->stream>>>(cnt, N, tile_start, total_items, (int64_t*)((char*)h->pinned_dev + 64),
                                                       h->engine_split ? 32 : kChains, rec_start);
        RNNWF_HIP(h, hipGetLastError());
    }
    a.sampling = 0;
    a.tile_start = tile_start;
    a.cnt = cnt;
    a.items = (const SwapItem*)h->tiles.p;
    a.cap = cap;
    a.contrib = (double2*)h->lpq.p;
    const int64_t max_tiles = (int64_t)N * ((2 * ns + kChains - 1) / kChains + 2);
    if (stack) {
        a.rec_start = rec_start;
        if (int rc = crnn_stack_swap(h, a, max_tiles, stack_max_records(N, ns))) return rc;
    } else if (h->engine_split) {
        if (int rc = crnn_split_swap(h, a, max_tiles)) return rc;
    } else {
        if (int rc = launch_swap(h, a, max_tiles)) return rc;
    }
    {
        TimedLaunch tl(h, 2);
        j1j2_eloc_kernel<<<(unsigned)((ns + 255) / 256), 256, 0, h->stream>>>((const double2*)h->lpq.p, diag, ns, N,
                                                                            (float2*)h->eloc.p, N <= 256 ? cnt : nullptr);
        RNNWF_HIP(h, hipGetLastError());
        h->j1j2_cnt_clean = N <= 256;
    }
    return 0;
}

// after a stream sync: number of scored configurations (the reference's len_sigmas) and work counters
int64_t collect_totals(rnnwf_handle* h, int64_t ns) {
    const int64_t* t = (const int64_t*)((char*)h->pinned + 64);
    h->work[0] += (double)t[1];
    h->work[1] += (double)t[2] * (h->engine_split ? (h->NL > 1 ? stack_split_flops_per_step(h) : crnn_split_flops_per_step(h))
                                                  : (double)(3 * h->NFULL + 1) * (4 * h->NFULL + 1) * 2048.0);
    return t[0] + ns;   // + one diagonal configuration per sample
}

}  // namespace

int rnnwf::crnn_pack_image(rnnwf_handle* h, std::vector<char>& img) {
    // swap-pass engine: bf16x3 on the matrix core (RNNWF_ENGINE=f32: f32-input MFMA everywhere; above 68 units the w3
    // fragments are read through L2, split_stream.hip)
    // stacked layers of 37..50 units: a pipeline of bf16x3 kernels, one per layer (split.hip: crnn_stack_swap); other stacks and
    // > 100 units: f32-input MFMA
    h->engine_split = h->knobs.engine != 1 && (h->NL == 1 ? h->NFULL <= 6 : stack_split_available(h));
    if (h->engine_split && h->NL > 1) {
        if (int rc = crnn_stack_pack(h)) return rc;
    } else if (h->engine_split) {
        std::vector<char> simg;
        if (int rc = crnn_split_pack(h, simg)) return rc;
        if (int rc = ensure(h, h->wsplit, simg.size())) return rc;
        if (int rc = upload(h, h->wsplit.p, simg.data(), simg.size())) return rc;
    }
    if (int rc = base_bf_pack(h)) return rc;
    CRNN_DISPATCH(h, { img = K::pack(h); return 0; });
    return h->fail(RNNWF_ERR_INVALID, "no cRNN kernel for NFULL=%d", h->NFULL);
}

int rnnwf::crnn_sample(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset, int32_t* out,
                       double* out_log) {
    const int W = (h->N + 31) / 32;
    h->last_ns = 0;
    if (int rc = ensure(h, h->bits, (size_t)W * ns * 4)) return rc;
    if (int rc = ensure(h, h->out_lp, (size_t)ns * 8)) return rc;
    CrnnArgs a = base_args(h, ns);
    a.bits = (uint32_t*)h->bits.p;
    a.out_logp = (double*)h->out_lp.p;
    a.sampling = 1;
    a.seed = seed; a.step = step; a.sample_offset = offset;
    if (int rc = launch_base(h, a)) return rc;
    if (int rc = unpack_and_download(h, h->bits, ns, out, nullptr)) return rc;
    if (out_log) RNNWF_HIP(h, hipMemcpyAsync(out_log, h->out_lp.p, (size_t)ns * 8, hipMemcpyDeviceToHost, h->stream));
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    return RNNWF_OK;
}

int rnnwf::crnn_log_amp(rnnwf_handle* h, const int32_t* samples, int64_t B, float* out_re_im, double* out_logp) {
    const int N = h->N;
    h->last_ns = 0;
    for (int64_t off = 0; off < B; off += kChunk) {
        const int64_t nb = std::min(kChunk, B - off);
        if (int rc = upload_and_pack(h, samples + off * N, nb, h->bits, 0, nullptr)) return rc;
        if (int rc = ensure(h, h->camp, (size_t)nb * sizeof(float2))) return rc;
        if (int rc = ensure(h, h->out_lp, (size_t)nb * 8)) return rc;
        CrnnArgs a = base_args(h, nb);
        a.bits = (uint32_t*)h->bits.p;
        a.out_amp = (float2*)h->camp.p;
        a.out_logp = (double*)h->out_lp.p;
        if (int rc = launch_base(h, a)) return rc;
        if (out_re_im)
            RNNWF_HIP(h, hipMemcpyAsync(out_re_im + 2 * off, h->camp.p, (size_t)nb * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
        if (out_logp)
            RNNWF_HIP(h, hipMemcpyAsync(out_logp + off, h->out_lp.p, (size_t)nb * 8, hipMemcpyDeviceToHost, h->stream));
        RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    }
    return RNNWF_OK;
}

int rnnwf::crnn_j1j2_eloc(rnnwf_handle* h, const int32_t* samples, int64_t ns, const double* J1, const double* J2,
                          const double* Bz, int periodic, int marshall, float* eloc, int64_t* ncon) {
    const int N = h->N;
    h->last_ns = 0;
    {
        std::vector<double> c((size_t)3 * N);
        std::copy(J1, J1 + N, c.begin());
        std::copy(J2, J2 + N, c.begin() + N);
        std::copy(Bz, Bz + N, c.begin() + 2 * N);
        if (int rc = upload_couplings(h, c.data(), c.size())) return rc;
    }
    const int64_t chunk = max_chains_per_pass(h);
    int64_t total = 0;
    for (int64_t off = 0; off < ns; off += chunk) {
        const int64_t nb = std::min(chunk, ns - off);
        if (int rc = upload_and_pack(h, samples + off * N, nb, h->bits, 0, nullptr)) return rc;
        if (int rc = j1j2_on_device(h, nb, false, 0, 0, 0, (const double*)h->coupl.p, periodic, marshall)) return rc;
        RNNWF_HIP(h, hipMemcpyAsync(eloc + 2 * off, h->eloc.p, (size_t)nb * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
        RNNWF_HIP(h, hipStreamSynchronize(h->stream));
        total += collect_totals(h, nb);
    }
    if (ncon) *ncon = total;
    return RNNWF_OK;
}

int rnnwf::crnn_load_batch(rnnwf_handle* h, const int32_t* samples, int64_t ns) {
    const int N = h->N;
    if (ns > max_chains_per_pass(h))
        return h->fail(RNNWF_ERR_NOMEM, "rnnwf_load_batch: %lld samples exceed the checkpoint budget; split the batch", (long long)ns);
    const std::vector<double> zeros((size_t)3 * N, 0.0);              // no bonds: base pass + checkpoints only
    if (int rc = upload_couplings(h, zeros.data(), zeros.size())) return rc;
    if (int rc = upload_and_pack(h, samples, ns, h->bits, 0, nullptr)) return rc;
    return j1j2_on_device(h, ns, false, 0, 0, 0, (const double*)h->coupl.p, 0, 0);
}

int rnnwf::crnn_vmc_step(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset,
                         const double* couplings, int32_t* out_samples, float* out_eloc, double* moments) {
    const int N = h->N;
    const int W = (N + 31) / 32;
    if (ns > max_chains_per_pass(h))
        return h->fail(RNNWF_ERR_NOMEM, "rnnwf_vmc_step: %lld samples exceed the checkpoint budget; split the batch",
                       (long long)ns);
    if (int rc = ensure(h, h->bits, (size_t)W * ns * 4)) return rc;
    if (int rc = upload_couplings(h, couplings, (size_t)3 * N)) return rc;
    const int periodic = couplings[3 * N] != 0.0, marshall = couplings[3 * N + 1] != 0.0;
    if (int rc = j1j2_on_device(h, ns, true, seed, step, offset, (const double*)h->coupl.p, periodic, marshall)) return rc;
    if (out_samples) if (int rc = unpack_and_download(h, h->bits, ns, out_samples, nullptr)) return rc;
    if (out_eloc) RNNWF_HIP(h, hipMemcpyAsync(out_eloc, h->eloc.p, (size_t)ns * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
    h->last_ns = ns;                  // bits, hck and eloc stay resident for rnnwf_vmc_gradient
    h->last_has_ckpt = true;
    if (int rc = run_moments(h, h->eloc.p, ns, true, moments)) return rc;   // syncs the stream (moments == nullptr, device-resident training: it does not)
    if (moments) collect_totals(h, ns);
    return RNNWF_OK;
}
