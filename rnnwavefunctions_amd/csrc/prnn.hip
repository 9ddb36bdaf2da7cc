// prnn.hip - host side of the positive GRU RNN wave function (models GRU1D, GRU1D_PARITY, GRU1D_F64):
// sample / log_probability / fused TFIM local energies / fused VMC step.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gru_kernels.h"
#include "ml_kernels.h"
#include "models.h"
#include "pack.h"

using namespace rnnwf;

namespace {

constexpr size_t kHckBudget = (size_t)48 << 30;  // bytes of hidden-state checkpoints per pass
constexpr int64_t kLogProbChunk = (int64_t)1 << 20;

template <typename T, int NFULL, int WAVES>
struct Launch {
    using L = GruLayout<T, NFULL, 1>;
    static int blocks_per_cu(rnnwf_handle* h, const void* fn, int* out) { return rnnwf::blocks_per_cu(h, fn, WAVES * 64, L::LDS_BYTES, out); }
    // fewer 16-chain blocks than SIMDs: the cooperative kernel (NFULL + 1 waves per block) cuts the per-site latency
    static int base_coop(rnnwf_handle* h, const PrnnArgs& a) {
        if constexpr (std::is_same<T, float>::value && NFULL <= 4) {
            const void* fn = (const void*)prnn_base_coop_kernel<NFULL>;
            const size_t lds = L::BYTES + (size_t)2 * L::KT * 64 * 4 + 2 * 64 * 4;
            int bpc = 0;
            if (int rc = rnnwf::blocks_per_cu(h, fn, (NFULL + 1) * 64, lds, &bpc)) return rc;
            const unsigned grid = (unsigned)std::min<int64_t>(a.nsb, (int64_t)bpc * h->cu_count);
            TimedLaunch tl(h, 0);
            prnn_base_coop_kernel<NFULL><<<grid, (NFULL + 1) * 64, lds, h->stream>>>(a);
            RNNWF_HIP(h, hipGetLastError());
        }
        return 0;
    }
    static int base(rnnwf_handle* h, const PrnnArgs& a) {
        // f32 models of 37..52 units: the cooperative kernel on the bf16 matrix core, for every batch size (a batch and its
        // shards always take the same kernel); RNNWF_BASE=f32 / RNNWF_NO_COOP=1 keep the f32-input-MFMA kernels
        if (std::is_same<T, float>::value && NFULL <= 3 && base_bf_available(h)) return prnn_base_coop_bf(h, a);
        if (std::is_same<T, float>::value && NFULL <= 4 && a.nsb <= (int64_t)4 * h->cu_count && !h->knobs.no_coop)
            return base_coop(h, a);
        const void* fn = (const void*)prnn_base_kernel<T, NFULL, WAVES>;
        int bpc = 0;
        if (int rc = blocks_per_cu(h, fn, &bpc)) return rc;
        const int64_t need = (a.nsb + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
        TimedLaunch tl(h, 0);
        prnn_base_kernel<T, NFULL, WAVES><<<grid, WAVES * 64, L::LDS_BYTES, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static int flip(rnnwf_handle* h, const PrnnArgs& a) {
        const void* fn = (const void*)prnn_flip_kernel<T, NFULL, WAVES>;
        int bpc = 0;
        if (int rc = blocks_per_cu(h, fn, &bpc)) return rc;
        const int64_t need = (a.ntiles + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
        TimedLaunch tl(h, 1);
        prnn_flip_kernel<T, NFULL, WAVES><<<grid, WAVES * 64, L::LDS_BYTES, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static std::vector<char> pack(const rnnwf_handle* h) { return pack_gru_image<T, NFULL, 1>(h); }
    static size_t hck_bytes_per_block() { return (size_t)L::KT * 64 * sizeof(T); }
    static double mfma_flops_per_step() { return (double)L::NT * L::KT * 2048.0; }
};

// stacked layers: same interface, kernels of ml_kernels.h
template <typename T, int NFULL, int NL, int WAVES>
struct MLaunchL {
    using M = MlCore<NFULL, NL, T>;
    static int blocks_per_cu(rnnwf_handle* h, const void* fn, int* out) { return rnnwf::blocks_per_cu(h, fn, WAVES * 64, M::BYTES, out); }
    static int base(rnnwf_handle* h, const PrnnArgs& a) {
        // all layers' images resident in LDS (f32, up to 52 units): the gate tiles of every layer spread over NFULL + 1 waves per block
        // of 16 chains (ml_coop.h) - for every batch size, its accumulation order is not the one-wave kernel's; RNNWF_NO_COOP=1 keeps that one
        if constexpr (std::is_same<T, float>::value && MlCoopLayout<NFULL, NL, 1>::FITS && M::SPILL == 0) {
            if (!h->knobs.no_coop) {
                using ML = MlCoopLayout<NFULL, NL, 1>;
                const void* cfn = (const void*)prnn_base_coop_kernel<NFULL, false, NL>;
                int cb = 0;
                if (int rc = rnnwf::blocks_per_cu(h, cfn, ML::THREADS, ML::LDS, &cb)) return rc;
                const int64_t need = (a.nsb + ML::NB - 1) / ML::NB;
                const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(need, (int64_t)cb * h->cu_count));
                TimedLaunch tl(h, 0);
                prnn_base_coop_kernel<NFULL, false, NL><<<grid, ML::THREADS, ML::LDS, h->stream>>>(a);
                RNNWF_HIP(h, hipGetLastError());
                return 0;
            }
        }
        const void* fn = (const void*)prnn_ml_base_kernel<T, NFULL, NL, WAVES>;
        int bpc = 0;
        if (int rc = blocks_per_cu(h, fn, &bpc)) return rc;
        // waves per workgroup: as few as still cover the batch with every resident workgroup busy (config 2 with two layers: 625
        // blocks of 16 chains -> 3 waves on each of 209 CUs instead of 8 on 79; measured 1.26 -> see DESIGN.md)
        const int64_t slots = (int64_t)bpc * h->cu_count;
        const int wpb = (int)std::max<int64_t>(1, std::min<int64_t>(WAVES, (a.nsb + slots - 1) / slots));
        const int64_t need = (a.nsb + wpb - 1) / wpb;
        const unsigned grid = (unsigned)std::min<int64_t>(need, slots);
        TimedLaunch tl(h, 0);
        prnn_ml_base_kernel<T, NFULL, NL, WAVES><<<grid, wpb * 64, M::BYTES, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static int flip(rnnwf_handle* h, const PrnnArgs& a) {
        const void* fn = (const void*)prnn_ml_flip_kernel<T, NFULL, NL, WAVES>;
        int bpc = 0;
        if (int rc = blocks_per_cu(h, fn, &bpc)) return rc;
        const int64_t need = (a.ntiles + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
        TimedLaunch tl(h, 1);
        prnn_ml_flip_kernel<T, NFULL, NL, WAVES><<<grid, WAVES * 64, M::BYTES, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static std::vector<char> pack(const rnnwf_handle* h) {
        std::vector<char> img = pack_gru_image<T, NFULL, 1>(h);
        for (int l = 1; l < NL; ++l) {
            const std::vector<char> up = pack_upper_image<NFULL, T>(h, l);
            img.insert(img.end(), up.begin(), up.end());
        }
        return img;
    }
    static size_t hck_bytes_per_block() { return (size_t)NL * M::KT * 64 * sizeof(T); }
    static double mfma_flops_per_step() {
        return ((double)M::C0::L::NT + 2.0 * (NL - 1) * M::CU::U::NT) * M::KT * 2048.0;
    }
};

// one place that maps (dtype, NFULL, layers) to an instantiation
#define PRNN_DISPATCH(h, EXPR)                                                              \
    do {                                                                                    \
        if ((h)->NL == 2 && (h)->f64) {                                                     \
            switch ((h)->NFULL) {                                                           \
                case 1: { using K = MLaunchL<double, 1, 2, 4>; EXPR; }                      \
                case 2: { using K = MLaunchL<double, 2, 2, 8>; EXPR; }                      \
                case 3: { using K = MLaunchL<double, 3, 2, 4>; EXPR; }                      \
                case 4: { using K = MLaunchL<double, 4, 2, 4>; EXPR; }                      \
            }                                                                               \
        } else if ((h)->NL == 3 && (h)->f64) {                                              \
            switch ((h)->NFULL) {                                                           \
                case 1: { using K = MLaunchL<double, 1, 3, 4>; EXPR; }                      \
                case 2: { using K = MLaunchL<double, 2, 3, 4>; EXPR; }                      \
                case 3: { using K = MLaunchL<double, 3, 3, 4>; EXPR; }                      \
                case 4: { using K = MLaunchL<double, 4, 3, 4>; EXPR; }                      \
            }                                                                               \
        } else if ((h)->NL == 4 && (h)->f64) {                                              \
            switch ((h)->NFULL) {                                                           \
                case 1: { using K = MLaunchL<double, 1, 4, 4>; EXPR; }                      \
                case 2: { using K = MLaunchL<double, 2, 4, 4>; EXPR; }                      \
                case 3: { using K = MLaunchL<double, 3, 4, 4>; EXPR; }                      \
                case 4: { using K = MLaunchL<double, 4, 4, 4>; EXPR; }                      \
            }                                                                               \
        } else if ((h)->NL == 4) {                                                          \
            switch ((h)->NFULL) {                                                           \
                case 1: { using K = MLaunchL<float, 1, 4, 4>; EXPR; }                       \
                case 2: { using K = MLaunchL<float, 2, 4, 4>; EXPR; }                       \
                case 3: { using K = MLaunchL<float, 3, 4, 4>; EXPR; }                       \
                case 4: { using K = MLaunchL<float, 4, 4, 4>; EXPR; }                       \
                case 6: { using K = MLaunchL<float, 6, 4, 4>; EXPR; }                       \
            }                                                                               \
        } else if ((h)->NL == 2) {                                                          \
            switch ((h)->NFULL) {                                                           \
                case 1: { using K = MLaunchL<float, 1, 2, 4>; EXPR; }                       \
                case 2: { using K = MLaunchL<float, 2, 2, 4>; EXPR; }                       \
                case 3: { using K = MLaunchL<float, 3, 2, 8>; EXPR; }                       \
                case 4: { using K = MLaunchL<float, 4, 2, 4>; EXPR; }                       \
                case 6: { using K = MLaunchL<float, 6, 2, 4>; EXPR; }                       \
            }                                                                               \
        } else if ((h)->NL == 3) {                                                          \
            switch ((h)->NFULL) {                                                           \
                case 1: { using K = MLaunchL<float, 1, 3, 4>; EXPR; }                       \
                case 2: { using K = MLaunchL<float, 2, 3, 8>; EXPR; }                       \
                case 3: { using K = MLaunchL<float, 3, 3, 8>; EXPR; }                       \
                case 4: { using K = MLaunchL<float, 4, 3, 4>; EXPR; }                       \
                case 6: { using K = MLaunchL<float, 6, 3, 4>; EXPR; }                       \
            }                                                                               \
        } else if (!(h)->f64) {                                                             \
            switch ((h)->NFULL) {                                                           \
                case 1: { using K = Launch<float, 1, 4>; EXPR; }                            \
                case 2: { using K = Launch<float, 2, 4>; EXPR; }                            \
                case 3: { using K = Launch<float, 3, 4>; EXPR; }                            \
                case 4: { using K = Launch<float, 4, 4>; EXPR; }                            \
                case 6: { using K = Launch<float, 6, 8>; EXPR; }                            \
                case 8: { using K = Launch<float, 8, 4>; EXPR; }                            \
                case 12: { using K = Launch<float, 12, 4>; EXPR; }                          \
                case 16: { using K = Launch<float, 16, 4>; EXPR; }                          \
            }                                                                               \
        } else {                                                                            \
            switch ((h)->NFULL) {                                                           \
                case 1: { using K = Launch<double, 1, 4>; EXPR; }                           \
                case 2: { using K = Launch<double, 2, 4>; EXPR; }                           \
                case 3: { using K = Launch<double, 3, 4>; EXPR; }                           \
                case 4: { using K = Launch<double, 4, 8>; EXPR; }                           \
                case 6: { using K = Launch<double, 6, 4>; EXPR; }                           \
            }                                                                               \
        }                                                                                   \
    } while (0)

int launch_base(rnnwf_handle* h, const PrnnArgs& a) {
    PRNN_DISPATCH(h, return K::base(h, a));
    return h->fail(RNNWF_ERR_INVALID, "no pRNN kernel for NFULL=%d f64=%d", h->NFULL, (int)h->f64);
}
int launch_flip(rnnwf_handle* h, const PrnnArgs& a) {
    PRNN_DISPATCH(h, return K::flip(h, a));
    return h->fail(RNNWF_ERR_INVALID, "no pRNN kernel for NFULL=%d f64=%d", h->NFULL, (int)h->f64);
}
size_t hck_bytes_per_block(rnnwf_handle* h) {
    PRNN_DISPATCH(h, return K::hck_bytes_per_block());
    return 0;
}
double mfma_flops_per_step(rnnwf_handle* h) {
    PRNN_DISPATCH(h, return K::mfma_flops_per_step());
    return 0;
}

PrnnArgs base_args(rnnwf_handle* h, int64_t ns) {
    PrnnArgs a{};
    a.wimg = h->wimg.p;
    a.N = h->N;
    a.ns = ns;
    a.nsb = (ns + kChains - 1) / kChains;
    a.ablate = h->knobs.ablate_base & (8 | 16 | 32);   // 0 unless a -DRNNWF_DIAGNOSTICS build read RNNWF_ABLATE_BASE
    return a;
}

// row_of_pos map for the reversed pass of the parity-symmetric model: position n of the reversed chain
// is site N-1-n, so its flip lands in lpq row (N-1-n)+1.
int ensure_reverse_map(rnnwf_handle* h) {
    const int N = h->N;
    if (h->maps.p) return 0;
    std::vector<int32_t> m(N);
    for (int n = 0; n < N; ++n) m[n] = N - n;
    if (int rc = ensure(h, h->maps, (size_t)N * 4)) return rc;
    RNNWF_HIP(h, hipMemcpy(h->maps.p, m.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    return 0;
}

// bf16x3 or f32-input MFMA for this flip pass?  The bf16x3 kernel works on 32-chain tiles; when there are not enough
// of them for two waves per SIMD (config 1: 304 tiles for 1 024 SIMDs) the 16-chain f32 kernel fills the chip better
// (measured 0.0198 vs 0.0223 ms at config 1).  RNNWF_ENGINE=bf16x3 pins the bf16x3 engine.
bool use_split(rnnwf_handle* h, int64_t ns_pass) {
    const int64_t ns = std::max<int64_t>(h->call_ns, ns_pass);     // the whole call decides, not the pass
    const int64_t tiles32 = (int64_t)(h->N - 1) * ((ns + 31) / 32);
    const bool split = h->engine_split && (h->engine_forced || tiles32 >= (int64_t)8 * h->cu_count);
    h->last_flip_engine = split ? 1 : 0;
    return split;
}

// The flip pass of one direction on the engine the call chose, with its work counters (cell evaluations, MFMA flops issued).
int flip_pass(rnnwf_handle* h, const PrnnArgs& a, int64_t ns) {
    const int N = h->N;
    const double wave_steps32 = (double)((ns + 31) / 32) * N * (N - 1) / 2.0;
    if (use_split(h, ns)) {
        if (h->NL > 1) {
            if (int rc = prnn_stack_flip(h, a)) return rc;
            h->work[1] += wave_steps32 * stack_split_flops_per_step(h);
        } else {
            if (int rc = prnn_split_flip(h, a)) return rc;
            h->work[1] += wave_steps32 * prnn_split_flops_per_step(h);
        }
    } else {
        if (int rc = launch_flip(h, a)) return rc;
        h->work[1] += (double)a.nsb * N * (N - 1) / 2.0 * mfma_flops_per_step(h);
    }
    h->work[0] += (double)ns * N * (N - 1) / 2.0;
    return 0;
}

// Fused local energies of ns chains whose packed spins are already in h->bits (and, for the parity
// model, reversed in h->bits2): base pass with checkpoints -> flip pass -> assembly.  Leaves E_loc in
// h->eloc and the log-prob queue in h->lpq.
int eloc_on_device(rnnwf_handle* h, int64_t ns, bool sampling, uint64_t seed, uint64_t step, int64_t offset,
                   int Nx, int Ny, const double* Jz_dev, double Bx) {
    const int N = h->N;
    const int64_t nsb = (ns + kChains - 1) / kChains;
    const bool parity = h->model == RNNWF_MODEL_GRU1D_PARITY;
    const size_t hck_bytes = (size_t)(h->NL > 1 ? N : std::max(N - 1, 1)) * nsb * hck_bytes_per_block(h);
    if (int rc = ensure(h, h->lpq, (size_t)(N + 1) * ns * 8)) return rc;
    if (int rc = ensure(h, h->eloc, (size_t)ns * 8)) return rc;
    if (int rc = ensure(h, h->hck, hck_bytes)) return rc;    // always: the gradient pass reuses the states

    PrnnArgs a = base_args(h, ns);
    a.bits = (uint32_t*)h->bits.p;
    a.hck = h->hck.p;
    a.lpq = (double*)h->lpq.p;
    a.sampling = sampling ? 1 : 0;
    a.seed = seed; a.step = step; a.sample_offset = offset;
    if (int rc = launch_base(h, a)) return rc;
    if (Bx != 0.0 && N > 1) {
        a.ntiles = (int64_t)(N - 1) * nsb;
        a.sampling = 0;
        a.ablate |= h->knobs.ablate & 15;   // 0 unless a -DRNNWF_DIAGNOSTICS build read RNNWF_ABLATE
        if (int rc = flip_pass(h, a, ns)) return rc;
    }
    if (parity) {
        // second direction on the reversed chains, then log(0.5 (e^a + e^b)) row by row
        if (sampling) {  // reversed bits from the freshly drawn spins
            if (int rc = unpack_device(h, h->bits, ns, nullptr)) return rc;
            if (int rc = pack_device(h, ns, h->bits2, 1, nullptr)) return rc;
        }
        if (int rc = ensure_reverse_map(h)) return rc;
        if (int rc = ensure(h, h->lpq2, (size_t)(N + 1) * ns * 8)) return rc;
        PrnnArgs b = base_args(h, ns);
        b.bits = (uint32_t*)h->bits2.p;
        b.hck = a.hck;
        b.lpq = (double*)h->lpq2.p;
        b.row_of_pos = (const int32_t*)h->maps.p;
        if (int rc = launch_base(h, b)) return rc;
        if (Bx != 0.0 && N > 1) {
            b.ntiles = (int64_t)(N - 1) * nsb;
            if (int rc = flip_pass(h, b, ns)) return rc;
        }
        if (int rc = run_parity_combine(h, (const double*)h->lpq.p, (const double*)h->lpq2.p, (int64_t)(N + 1) * ns,
                                        (double*)h->lpq.p)) return rc;
    }
    return run_tfim_eloc(h, (const uint32_t*)h->bits.p, (const double*)h->lpq.p, ns, Nx, Ny, nullptr, Jz_dev, Bx,
                         (double*)h->eloc.p);
}

int64_t max_chains_per_pass(rnnwf_handle* h) {
    size_t per_block = (size_t)(h->NL > 1 ? h->N : std::max(h->N - 1, 1)) * hck_bytes_per_block(h);
    if (h->NL > 1 && h->engine_split)                         // per 16 chains: half a 32-chain tile column of the layer pipeline's records
        per_block += stack_record_bytes_per_32_chains(h, (int64_t)h->N * (h->N - 1) / 2) / 2;
    const int64_t blocks = std::max<int64_t>(1, (int64_t)(state_budget_bytes(h, kHckBudget) / per_block));
    return blocks * kChains;
}

}  // namespace

// Teacher-forced base pass with checkpoints over the resident spins (h->bits, or the reversed ones in h->bits2), log P of every
// chain to out_lp: what a backward pass of the parity-symmetric model needs per direction (grad.hip).
int rnnwf::prnn_teacher_base(rnnwf_handle* h, int64_t ns, bool reversed, double* out_lp) {
    PrnnArgs a = base_args(h, ns);
    a.bits = (uint32_t*)(reversed ? h->bits2.p : h->bits.p);
    a.hck = h->hck.p;
    a.out_lp = out_lp;
    return launch_base(h, a);
}

int rnnwf::prnn_pack_image(rnnwf_handle* h, std::vector<char>& img) {
    // flip-pass engine: bf16x3 on the matrix core for the f32 models (RNNWF_ENGINE=f32 keeps the f32-input MFMA
    // everywhere; above 68 units the w3 fragments of the image are read through L2, split_stream.hip); the base pass, sampling and log_probability always run the f32-MFMA kernels
    // stacked layers: 37..50 units run as a pipeline of bf16x3 kernels, one per layer (split.hip: prnn_stack_flip); other widths
    // keep the f32-input MFMA
    h->engine_split = !h->f64 && h->knobs.engine != 1 && (h->NL == 1 ? h->NFULL <= 6 : stack_split_available(h));     // above 100 units: f32-input MFMA, image through L2
    h->engine_forced = h->knobs.engine >= 2;
    if (h->engine_split && h->NL > 1) {
        if (int rc = prnn_stack_pack(h)) return rc;
    } else if (h->engine_split) {
        std::vector<char> simg;
        if (int rc = prnn_split_pack(h, simg)) return rc;
        if (int rc = ensure(h, h->wsplit, simg.size())) return rc;
        if (int rc = upload(h, h->wsplit.p, simg.data(), simg.size())) return rc;
    }
    if (int rc = base_bf_pack(h)) return rc;
    PRNN_DISPATCH(h, { img = K::pack(h); return 0; });
    return h->fail(RNNWF_ERR_INVALID, "no pRNN kernel for NFULL=%d f64=%d", h->NFULL, (int)h->f64);
}

int rnnwf::prnn_log_prob(rnnwf_handle* h, const int32_t* samples, int64_t B, double* out) {
    const int N = h->N;
    h->last_ns = 0;
    const bool parity = h->model == RNNWF_MODEL_GRU1D_PARITY;
    for (int64_t off = 0; off < B; off += kLogProbChunk) {
        const int64_t nb = std::min(kLogProbChunk, B - off);
        if (int rc = upload_and_pack(h, samples + off * N, nb, h->bits, 0, nullptr)) return rc;
        if (int rc = ensure(h, h->out_lp, (size_t)nb * 8)) return rc;
        PrnnArgs a = base_args(h, nb);
        a.bits = (uint32_t*)h->bits.p;
        a.out_lp = (double*)h->out_lp.p;
        if (int rc = launch_base(h, a)) return rc;
        if (parity) {
            if (int rc = pack_device(h, nb, h->bits2, 1, nullptr)) return rc;
            if (int rc = ensure(h, h->out_lp2, (size_t)nb * 8)) return rc;
            a.bits = (uint32_t*)h->bits2.p;
            a.out_lp = (double*)h->out_lp2.p;
            if (int rc = launch_base(h, a)) return rc;
            if (int rc = run_parity_combine(h, (const double*)h->out_lp.p, (const double*)h->out_lp2.p, nb,
                                            (double*)h->out_lp.p)) return rc;
        }
        RNNWF_HIP(h, hipMemcpyAsync(out + off, h->out_lp.p, (size_t)nb * 8, hipMemcpyDeviceToHost, h->stream));
        RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    }
    return RNNWF_OK;
}

int rnnwf::prnn_sample(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset, int32_t* out,
                       double* out_log) {
    const int N = h->N;
    const int W = (N + 31) / 32;
    h->last_ns = 0;
    if (int rc = ensure(h, h->bits, (size_t)W * ns * 4)) return rc;
    if (int rc = ensure(h, h->out_lp, (size_t)ns * 8)) return rc;
    PrnnArgs a = base_args(h, ns);
    a.bits = (uint32_t*)h->bits.p;
    a.out_lp = (double*)h->out_lp.p;
    a.sampling = 1;
    a.seed = seed; a.step = step; a.sample_offset = offset;
    if (int rc = launch_base(h, a)) return rc;
    if (int rc = unpack_and_download(h, h->bits, ns, out, nullptr)) return rc;
    if (out_log) {
        if (h->model == RNNWF_MODEL_GRU1D_PARITY) {  // symmetrised probability of the drawn configurations
            if (int rc = pack_device(h, ns, h->bits2, 1, nullptr)) return rc;
            if (int rc = ensure(h, h->out_lp2, (size_t)ns * 8)) return rc;
            PrnnArgs b = base_args(h, ns);
            b.bits = (uint32_t*)h->bits2.p;
            b.out_lp = (double*)h->out_lp2.p;
            if (int rc = launch_base(h, b)) return rc;
            if (int rc = run_parity_combine(h, (const double*)h->out_lp.p, (const double*)h->out_lp2.p, ns,
                                            (double*)h->out_lp.p)) return rc;
        }
        RNNWF_HIP(h, hipMemcpyAsync(out_log, h->out_lp.p, (size_t)ns * 8, hipMemcpyDeviceToHost, h->stream));
    }
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    return RNNWF_OK;
}

int rnnwf::prnn_tfim_eloc(rnnwf_handle* h, const int32_t* samples, int64_t ns, int Nx, int Ny, const double* Jz,
                          double Bx, double* eloc, double* log_probs) {
    const int N = h->N;
    h->last_ns = 0;
    h->call_ns = ns;
    if (int rc = upload_couplings(h, Jz, (size_t)N)) return rc;
    const int64_t chunk = max_chains_per_pass(h);
    for (int64_t off = 0; off < ns; off += chunk) {
        const int64_t nb = std::min(chunk, ns - off);
        if (int rc = upload_and_pack(h, samples + off * N, nb, h->bits, 0, nullptr)) return rc;
        if (h->model == RNNWF_MODEL_GRU1D_PARITY)
            if (int rc = pack_device(h, nb, h->bits2, 1, nullptr)) return rc;
        if (int rc = eloc_on_device(h, nb, false, 0, 0, 0, Nx, Ny, (const double*)h->coupl.p, Bx)) return rc;
        RNNWF_HIP(h, hipMemcpyAsync(eloc + off, h->eloc.p, (size_t)nb * 8, hipMemcpyDeviceToHost, h->stream));
        if (log_probs)
            RNNWF_HIP(h, hipMemcpy2DAsync(log_probs + off, (size_t)ns * 8, h->lpq.p, (size_t)nb * 8, (size_t)nb * 8,
                                          (size_t)N + 1, hipMemcpyDeviceToHost, h->stream));
        RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    }
    return RNNWF_OK;
}

// Teacher-forced base pass with checkpoints on caller-supplied samples (no flips, no energies): what the gradient needs
// resident besides E_loc, which rnnwf_load_batch uploads afterwards.
int rnnwf::prnn_load_batch(rnnwf_handle* h, const int32_t* samples, int64_t ns) {
    const int N = h->N;
    if (ns > max_chains_per_pass(h))
        return h->fail(RNNWF_ERR_NOMEM, "rnnwf_load_batch: %lld samples exceed the checkpoint budget; split the batch", (long long)ns);
    h->call_ns = ns;
    const std::vector<double> zeros((size_t)N, 0.0);
    if (int rc = upload_couplings(h, zeros.data(), (size_t)N)) return rc;
    if (int rc = upload_and_pack(h, samples, ns, h->bits, 0, nullptr)) return rc;
    if (h->model == RNNWF_MODEL_GRU1D_PARITY)
        if (int rc = pack_device(h, ns, h->bits2, 1, nullptr)) return rc;
    const int Nx = h->model == RNNWF_MODEL_GRU1D_F64 ? h->Nx : 1;
    const int Ny = h->model == RNNWF_MODEL_GRU1D_F64 ? h->Ny : N;
    return eloc_on_device(h, ns, false, 0, 0, 0, Nx, Ny, (const double*)h->coupl.p, 0.0);
}

int rnnwf::prnn_vmc_step(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset,
                         const double* couplings, int32_t* out_samples, double* out_eloc, double* moments) {
    const int N = h->N;
    const int W = (N + 31) / 32;
    if (ns > max_chains_per_pass(h))
        return h->fail(RNNWF_ERR_NOMEM, "rnnwf_vmc_step: %lld samples exceed the checkpoint budget; split the batch",
                       (long long)ns);
    h->call_ns = ns;
    if (int rc = ensure(h, h->bits, (size_t)W * ns * 4)) return rc;
    if (int rc = upload_couplings(h, couplings, (size_t)N)) return rc;
    const double Bx = couplings[N];
    const int Nx = h->model == RNNWF_MODEL_GRU1D_F64 ? h->Nx : 1;
    const int Ny = h->model == RNNWF_MODEL_GRU1D_F64 ? h->Ny : N;
    if (int rc = eloc_on_device(h, ns, true, seed, step, offset, Nx, Ny, (const double*)h->coupl.p, Bx)) return rc;
    if (out_samples) if (int rc = unpack_and_download(h, h->bits, ns, out_samples, nullptr)) return rc;
    if (out_eloc) RNNWF_HIP(h, hipMemcpyAsync(out_eloc, h->eloc.p, (size_t)ns * 8, hipMemcpyDeviceToHost, h->stream));
    h->last_ns = ns;              // bits, hck and eloc stay resident for rnnwf_vmc_gradient
    h->last_has_ckpt = true;
    return run_moments(h, h->eloc.p, ns, false, moments);
}
