// mdrnn.hip - host side of the 2D MDRNN wave function (model MDRNN2D, float64):
// sample / log_probability / fused 2D-TFIM local energies / fused VMC step.
#include <algorithm>

#include "grad_kernels.h"
#include "mdrnn_grad_kernels.h"
#include "tn_gemm.h"
#include "mdrnn_kernels.h"
#include "models.h"
#include "pack.h"

using namespace rnnwf;

namespace {

constexpr size_t kHsBudget = (size_t)24 << 30;   // bytes of per-site hidden states per pass

template <int NFULL, int WAVES>
struct MLaunch {
    using L = MdLayout<NFULL>;
    static int blocks_per_cu(rnnwf_handle* h, const void* fn, int* out) { return rnnwf::blocks_per_cu(h, fn, WAVES * 64, L::BYTES, out); }
    static int base(rnnwf_handle* h, const MdArgs& a) {
        const void* fn = (const void*)mdrnn_base_kernel<NFULL, WAVES>;
        int bpc = 0;
        if (int rc = blocks_per_cu(h, fn, &bpc)) return rc;
        const int64_t need = (a.nsb + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
        TimedLaunch tl(h, 0);
        mdrnn_base_kernel<NFULL, WAVES><<<grid, WAVES * 64, L::BYTES, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static constexpr size_t FLIP_LDS = L::BYTES + (size_t)WAVES * L::WORDS_BYTES;     // image + the waves' spin words
    static int flip_grid(rnnwf_handle* h, int64_t ntiles, unsigned* grid) {
        const void* fn = (const void*)mdrnn_flip_kernel<NFULL, WAVES>;
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, FLIP_LDS, &bpc)) return rc;
        const int64_t need = (ntiles + WAVES - 1) / WAVES;
        *grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
        return 0;
    }
#ifdef RNNWF_DIAGNOSTICS      // measured negative (profiles/r03_d_cfg4_prefetch.md), kept for A/B runs of tools/ only: not in the release library
    // prefetching variant (mdrnn_flip_pf_kernel): one 8-wave workgroup per CU, a staging slot per wave behind the spin words
    static constexpr int PF_WAVES = 8;
    static constexpr size_t PF_LDS = L::BYTES + (size_t)PF_WAVES * L::WORDS_BYTES + (size_t)PF_WAVES * ((L::KT + 1) / 2) * 1024;
    static constexpr bool PF_FITS = PF_LDS <= 160 * 1024;
    static int flip_pf(rnnwf_handle* h, MdArgs a) {
        if constexpr (PF_FITS) {
            const void* fn = (const void*)mdrnn_flip_pf_kernel<NFULL, PF_WAVES>;
            int bpc = 0;
            if (int rc = rnnwf::blocks_per_cu(h, fn, PF_WAVES * 64, PF_LDS, &bpc)) return rc;
            const int64_t need = (a.ntiles + PF_WAVES - 1) / PF_WAVES;
            const unsigned grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
            const size_t ring_bytes = (size_t)grid * PF_WAVES * a.Nx * ((L::KT + 1) / 2) * 64 * 16;
            if (int rc = ensure(h, h->rowbuf, ring_bytes)) return rc;
            a.ring = (double*)h->rowbuf.p;
            a.ablate = h->knobs.ablate;
            TimedLaunch tl(h, 1);
            mdrnn_flip_pf_kernel<NFULL, PF_WAVES><<<grid, PF_WAVES * 64, PF_LDS, h->stream>>>(a);
            RNNWF_HIP(h, hipGetLastError());
        }
        return 0;
    }
#endif
    static int flip(rnnwf_handle* h, MdArgs a) {
#ifdef RNNWF_DIAGNOSTICS
        if (PF_FITS && NFULL == 3 && h->knobs.md_prefetch) return flip_pf(h, a);      // A/B only: measured slower (profiles/r03_d_cfg4_prefetch.md)
#endif
        unsigned grid = 0;
        if (int rc = flip_grid(h, a.ntiles, &grid)) return rc;
        const size_t ring_bytes = (size_t)grid * WAVES * a.Nx * ((L::KT + 1) / 2) * 64 * 16;      // one slot per lattice column
        if (int rc = ensure(h, h->rowbuf, ring_bytes)) return rc;
        a.ring = (double*)h->rowbuf.p;
        a.ablate = h->knobs.ablate;   // 0 unless a -DRNNWF_DIAGNOSTICS build read RNNWF_ABLATE
        TimedLaunch tl(h, 1);
        mdrnn_flip_kernel<NFULL, WAVES><<<grid, WAVES * 64, FLIP_LDS, h->stream>>>(a);
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }
    static size_t hs_bytes_per_block() { return (size_t)((L::KT + 1) / 2) * 64 * 16; }
    static double mfma_flops_per_step() { return (double)NFULL * 2 * L::KT * 2048.0; }

    // S = double: the image of the committed parameters; S = Lin: the same code recording, for the device re-pack kernel of
    // train.hip, which parameters every element is made of (pack_value.h)
    template <class S = double>
    static std::vector<char> pack(const rnnwf_handle* h) {
        using Out = PackSink<S>;
        const int H = h->H;
        std::vector<char> img(L::BYTES, 0);
        Out::begin(img);
        const auto Wh = pvs<S>(h, "Wh_rnn_0");   // [H, H]
        const auto Uh = pvs<S>(h, "Uh_rnn_0");   // [2, H]
        const auto Wv = pvs<S>(h, "Wv_rnn_0");
        const auto Uv = pvs<S>(h, "Uv_rnn_0");
        const auto b = pvs<S>(h, "b_rnn_0");
        const auto Wd = pvs<S>(h, "wf_dense/kernel");
        const auto bd = pvs<S>(h, "wf_dense/bias");
        double* A = reinterpret_cast<double*>(img.data() + L::OFF_A);
        for (int t = 0; t < NFULL; ++t)
            for (int row = 0; row < 16; ++row) {
                const int unit = 16 * t + row;
                if (unit >= H) continue;
                for (int kq = 0; kq < 4; ++kq) {
                    const int lane = (kq << 4) | row;
                    for (int kk = 0; kk < 2 * L::KT; ++kk) {
                        const int k = 4 * (kk < L::KT ? kk : kk - L::KT) + kq;
                        if (k >= H) continue;
                        Out::put(&A[(((size_t)t * L::KT + kk / 2) * 64 + lane) * 2 + (kk & 1)],
                                 kk < L::KT ? Wh[(size_t)k * H + unit] : Wv[(size_t)k * H + unit]);
                    }
                }
            }
        double* WR = reinterpret_cast<double*>(img.data() + L::OFF_WR);
        for (int kk = 0; kk < 2 * L::KT; ++kk)
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * (kk < L::KT ? kk : kk - L::KT) + q;
                if (k >= H) continue;
                for (int j = 0; j < 4; ++j) {
                    const int unit = 16 * NFULL + j;
                    if (unit >= H) continue;
                    Out::put(&WR[(kk * 4 + q) * 4 + j], kk < L::KT ? Wh[(size_t)k * H + unit] : Wv[(size_t)k * H + unit]);
                }
            }
        // b + Uh[x_h] (BH), Uv[x_v] (BV) for x in {none, 0, 1}, and their nine sums (BHV): one accumulator start value per step
        auto bh_of = [&](int v, int unit) -> S { return b[unit] + (v ? Uh[(size_t)(v - 1) * H + unit] : S(0.0)); };
        auto bv_of = [&](int v, int unit) -> S { return v ? Uv[(size_t)(v - 1) * H + unit] : S(0.0); };
        for (int t = 0; t < L::NT; ++t)
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < 4; ++r) {
                    if (t == NFULL && r != 0) continue;
                    const int unit = t < NFULL ? 16 * t + 4 * r + q : 16 * NFULL + q;
                    if (unit >= H) continue;
                    const int k = t * 16 + q * 4 + r;
                    for (int v = 0; v < 3; ++v) {
                        Out::put(reinterpret_cast<double*>(img.data() + L::OFF_BH + v * L::SZ_B) + k, bh_of(v, unit));
                        Out::put(reinterpret_cast<double*>(img.data() + L::OFF_BV + v * L::SZ_B) + k, bv_of(v, unit));
                    }
                    for (int vh = 0; vh < 3; ++vh)
                        for (int vv = 0; vv < 3; ++vv)
                            Out::put(reinterpret_cast<double*>(img.data() + L::OFF_BHV + (size_t)(vh * 3 + vv) * L::SZ_B) + k,
                                     bh_of(vh, unit) + bv_of(vv, unit));
                }
        double* WD = reinterpret_cast<double*>(img.data() + L::OFF_WD);
        double* BD = reinterpret_cast<double*>(img.data() + L::OFF_BD);
        for (int kt = 0; kt < L::KT; ++kt)
            for (int q = 0; q < 4; ++q) {
                const int unit = 4 * kt + q;
                if (unit >= H) continue;
                Out::put(&WD[(kt * 4 + q) * 2], Wd[(size_t)unit * 2]);
                Out::put(&WD[(kt * 4 + q) * 2 + 1], Wd[(size_t)unit * 2 + 1]);
            }
        Out::put(&BD[0], bd[0]);
        Out::put(&BD[1], bd[1]);
        fill_f64_tables(reinterpret_cast<double*>(img.data() + L::OFF_TAB));
        double* WDD = reinterpret_cast<double*>(img.data() + L::OFF_WDD);
        for (int unit = 0; unit < H; ++unit) Out::put(&WDD[unit], Wd[(size_t)unit * 2 + 1] - Wd[(size_t)unit * 2]);   // slot 4 kt + q
        Out::put(&WDD[L::KT * 4], bd[1] - bd[0]);
        return img;
    }
};

#define MD_DISPATCH(h, EXPR)                                    \
    do {                                                        \
        switch ((h)->NFULL) {                                   \
            case 1: { using K = MLaunch<1, 4>; EXPR; }          \
            case 2: { using K = MLaunch<2, 4>; EXPR; }          \
            case 3: { using K = MLaunch<3, 4>; EXPR; }          \
            case 4: { using K = MLaunch<4, 4>; EXPR; }          \
            case 5: { using K = MLaunch<5, 4>; EXPR; }          \
        }                                                       \
    } while (0)

int launch_base(rnnwf_handle* h, const MdArgs& a) {
    MD_DISPATCH(h, return K::base(h, a));
    return h->fail(RNNWF_ERR_INVALID, "MDRNN: num_units > 84 is not implemented on gfx950 yet");
}
int launch_flip(rnnwf_handle* h, const MdArgs& a) {
    MD_DISPATCH(h, return K::flip(h, a));
    return h->fail(RNNWF_ERR_INVALID, "MDRNN: num_units > 84 is not implemented on gfx950 yet");
}
size_t hs_bytes_per_block(rnnwf_handle* h) {
    MD_DISPATCH(h, return K::hs_bytes_per_block());
    return 1;
}
double mfma_flops_per_step(rnnwf_handle* h) {
    MD_DISPATCH(h, return K::mfma_flops_per_step());
    return 0;
}

// device maps, 6 x N int32: col_of_pos | pos_of_site | row_of_pos | vert_pos | row_first | (spare)
struct Maps {
    const int32_t *col_of_pos, *pos_of_site, *row_of_pos, *vert_pos, *row_first;
};

int get_maps(rnnwf_handle* h, Maps* m) {
    const int Nx = h->Nx, Ny = h->Ny, N = h->N;
    if (!h->maps.p) {
        std::vector<int32_t> v((size_t)5 * N);
        auto pos_of = [&](int nx, int ny) { return ny * Nx + (ny % 2 == 0 ? nx : Nx - 1 - nx); };
        for (int p = 0; p < N; ++p) {
            const int ny = p / Nx, j = p % Nx;
            const int nx = ny % 2 == 0 ? j : Nx - 1 - j;
            const int k = nx * Ny + ny;                        // samples[b, nx, ny] in C order
            v[p] = k;
            v[(size_t)N + k] = p;
            v[(size_t)2 * N + p] = k + 1;                      // queue row of the flip at (nx, ny): nx*Ny + ny + 1
            v[(size_t)3 * N + p] = ny > 0 ? pos_of(nx, ny - 1) : -1;
            v[(size_t)4 * N + p] = j == 0 ? 1 : 0;
        }
        if (int rc = ensure(h, h->maps, v.size() * 4)) return rc;
        RNNWF_HIP(h, hipMemcpy(h->maps.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    }
    const int32_t* b = (const int32_t*)h->maps.p;
    m->col_of_pos = b;
    m->pos_of_site = b + N;
    m->row_of_pos = b + 2 * (size_t)N;
    m->vert_pos = b + 3 * (size_t)N;
    m->row_first = b + 4 * (size_t)N;
    return 0;
}

int64_t max_chains_per_pass(rnnwf_handle* h) {
    const size_t per_block = (size_t)h->N * hs_bytes_per_block(h);
    return std::max<int64_t>(1, (int64_t)(state_budget_bytes(h, kHsBudget) / per_block)) * kChains;
}

MdArgs base_args(rnnwf_handle* h, int64_t ns, const Maps& m) {
    MdArgs a{};
    a.wimg = h->wimg.p;
    a.N = h->N;
    a.Nx = h->Nx;
    a.rem = h->H - 16 * h->NFULL;
    a.ns = ns;
    a.nsb = (ns + kChains - 1) / kChains;
    a.vert_pos = m.vert_pos;
    a.row_first = m.row_first;
    a.row_of_pos = m.row_of_pos;
    return a;
}

int eloc_on_device(rnnwf_handle* h, int64_t ns, const Maps& m, bool sampling, uint64_t seed, uint64_t step,
                   int64_t offset, const double* Jz_dev, double Bx) {
    const int N = h->N;
    const int64_t nsb = (ns + kChains - 1) / kChains;
    if (int rc = ensure(h, h->hck, (size_t)N * nsb * hs_bytes_per_block(h))) return rc;
    if (int rc = ensure(h, h->lpq, (size_t)(N + 1) * ns * 8)) return rc;
    if (int rc = ensure(h, h->eloc, (size_t)ns * 8)) return rc;
    MdArgs a = base_args(h, ns, m);
    a.bits = (uint32_t*)h->bits.p;
    a.hs = (double*)h->hck.p;
    a.lpq = (double*)h->lpq.p;
    a.sampling = sampling ? 1 : 0;
    a.seed = seed; a.step = step; a.sample_offset = offset;
    if (int rc = launch_base(h, a)) return rc;
    if (Bx != 0.0 && N > 1) {
        a.sampling = 0;
        a.ntiles = (int64_t)(N - 1) * nsb;
        if (int rc = launch_flip(h, a)) return rc;
        h->work[0] += (double)ns * N * (N - 1) / 2.0;
        h->work[1] += (double)nsb * N * (N - 1) / 2.0 * mfma_flops_per_step(h);
    }
    return run_tfim_eloc(h, (const uint32_t*)h->bits.p, (const double*)h->lpq.p, ns, h->Nx, h->Ny, m.pos_of_site, Jz_dev,
                         Bx, (double*)h->eloc.p);
}

}  // namespace

int rnnwf::mdrnn_pack_image(rnnwf_handle* h, std::vector<char>& img) {
    if (h->N > 256) return h->fail(RNNWF_ERR_INVALID, "MDRNN: lattices above 256 sites are not implemented");
    MD_DISPATCH(h, { img = K::template pack<double>(h); return 0; });
    return h->fail(RNNWF_ERR_INVALID, "MDRNN: num_units > 84 is not implemented on gfx950 yet");
}

int rnnwf::mdrnn_log_prob(rnnwf_handle* h, const int32_t* samples, int64_t B, double* out) {
    const int N = h->N;
    h->last_ns = 0;
    Maps m;
    if (int rc = get_maps(h, &m)) return rc;
    const int64_t chunk = max_chains_per_pass(h);
    for (int64_t off = 0; off < B; off += chunk) {
        const int64_t nb = std::min(chunk, B - off);
        const int64_t nsb = (nb + kChains - 1) / kChains;
        if (int rc = upload_and_pack(h, samples + off * N, nb, h->bits, 0, m.col_of_pos)) return rc;
        if (int rc = ensure(h, h->out_lp, (size_t)nb * 8)) return rc;
        if (int rc = ensure(h, h->hck, (size_t)N * nsb * hs_bytes_per_block(h))) return rc;
        MdArgs a = base_args(h, nb, m);
        a.bits = (uint32_t*)h->bits.p;
        a.hs = (double*)h->hck.p;
        a.out_lp = (double*)h->out_lp.p;
        if (int rc = launch_base(h, a)) return rc;
        RNNWF_HIP(h, hipMemcpyAsync(out + off, h->out_lp.p, (size_t)nb * 8, hipMemcpyDeviceToHost, h->stream));
        RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    }
    return RNNWF_OK;
}

int rnnwf::mdrnn_sample(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset, int32_t* out,
                        double* out_log) {
    const int N = h->N;
    h->last_ns = 0;
    const int W = (N + 31) / 32;
    Maps m;
    if (int rc = get_maps(h, &m)) return rc;
    if (ns > max_chains_per_pass(h)) return h->fail(RNNWF_ERR_NOMEM, "rnnwf_sample: batch too large for one pass; split it");
    const int64_t nsb = (ns + kChains - 1) / kChains;
    if (int rc = ensure(h, h->bits, (size_t)W * ns * 4)) return rc;
    if (int rc = ensure(h, h->out_lp, (size_t)ns * 8)) return rc;
    if (int rc = ensure(h, h->hck, (size_t)N * nsb * hs_bytes_per_block(h))) return rc;
    MdArgs a = base_args(h, ns, m);
    a.bits = (uint32_t*)h->bits.p;
    a.hs = (double*)h->hck.p;
    a.out_lp = (double*)h->out_lp.p;
    a.sampling = 1;
    a.seed = seed; a.step = step; a.sample_offset = offset;
    if (int rc = launch_base(h, a)) return rc;
    if (int rc = unpack_and_download(h, h->bits, ns, out, m.pos_of_site)) return rc;
    if (out_log) RNNWF_HIP(h, hipMemcpyAsync(out_log, h->out_lp.p, (size_t)ns * 8, hipMemcpyDeviceToHost, h->stream));
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    return RNNWF_OK;
}

int rnnwf::mdrnn_tfim_eloc(rnnwf_handle* h, const int32_t* samples, int64_t ns, const double* Jz, double Bx,
                           double* eloc, double* log_probs) {
    const int N = h->N;
    h->last_ns = 0;
    Maps m;
    if (int rc = get_maps(h, &m)) return rc;
    if (int rc = upload_couplings(h, Jz, (size_t)N)) return rc;
    const int64_t chunk = max_chains_per_pass(h);
    for (int64_t off = 0; off < ns; off += chunk) {
        const int64_t nb = std::min(chunk, ns - off);
        if (int rc = upload_and_pack(h, samples + off * N, nb, h->bits, 0, m.col_of_pos)) return rc;
        if (int rc = eloc_on_device(h, nb, m, false, 0, 0, 0, (const double*)h->coupl.p, Bx)) return rc;
        RNNWF_HIP(h, hipMemcpyAsync(eloc + off, h->eloc.p, (size_t)nb * 8, hipMemcpyDeviceToHost, h->stream));
        if (log_probs)
            RNNWF_HIP(h, hipMemcpy2DAsync(log_probs + off, (size_t)ns * 8, h->lpq.p, (size_t)nb * 8, (size_t)nb * 8,
                                          (size_t)N + 1, hipMemcpyDeviceToHost, h->stream));
        RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    }
    return RNNWF_OK;
}

int rnnwf::mdrnn_load_batch(rnnwf_handle* h, const int32_t* samples, int64_t ns) {
    const int N = h->N;
    Maps m;
    if (int rc = get_maps(h, &m)) return rc;
    if (ns > max_chains_per_pass(h))
        return h->fail(RNNWF_ERR_NOMEM, "rnnwf_load_batch: %lld samples exceed the hidden-state budget; split the batch", (long long)ns);
    const std::vector<double> zeros((size_t)N, 0.0);
    if (int rc = upload_couplings(h, zeros.data(), (size_t)N)) return rc;
    if (int rc = upload_and_pack(h, samples, ns, h->bits, 0, m.col_of_pos)) return rc;
    return eloc_on_device(h, ns, m, false, 0, 0, 0, (const double*)h->coupl.p, 0.0);
}

int rnnwf::mdrnn_vmc_step(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset,
                          const double* couplings, int32_t* out_samples, double* out_eloc, double* moments) {
    const int N = h->N;
    const int W = (N + 31) / 32;
    Maps m;
    if (int rc = get_maps(h, &m)) return rc;
    if (ns > max_chains_per_pass(h))
        return h->fail(RNNWF_ERR_NOMEM, "rnnwf_vmc_step: %lld samples exceed the hidden-state budget; split the batch",
                       (long long)ns);
    if (int rc = ensure(h, h->bits, (size_t)W * ns * 4)) return rc;
    if (int rc = upload_couplings(h, couplings, (size_t)N)) return rc;
    if (int rc = eloc_on_device(h, ns, m, true, seed, step, offset, (const double*)h->coupl.p, couplings[N])) return rc;
    if (out_samples) if (int rc = unpack_and_download(h, h->bits, ns, out_samples, m.pos_of_site)) return rc;
    if (out_eloc) RNNWF_HIP(h, hipMemcpyAsync(out_eloc, h->eloc.p, (size_t)ns * 8, hipMemcpyDeviceToHost, h->stream));
    h->last_ns = ns;              // bits, hs and eloc stay resident for rnnwf_vmc_gradient
    h->last_has_ckpt = true;
    return run_moments(h, h->eloc.p, ns, false, moments);
}

// ---- gradient of the VMC cost (2DTFIM_2DRNN/Training2DRNN_2DTFIM.py:163-170) ----
namespace {

template <int NFULL, int WAVES>
struct MGrad {
    using G = MdGradLayout<NFULL>;

    template <class S = double>
    static std::vector<char> pack(const rnnwf_handle* h) {
        using Out = PackSink<S>;
        const int H = h->H;
        std::vector<char> img(G::BYTES, 0);
        Out::begin(img);
        const auto Wh = pvs<S>(h, "Wh_rnn_0");
        const auto Wv = pvs<S>(h, "Wv_rnn_0");
        const auto Wd = pvs<S>(h, "wf_dense/kernel");
        const auto bd = pvs<S>(h, "wf_dense/bias");
        double* A = reinterpret_cast<double*>(img.data() + G::OFF_A);
        for (int t = 0; t < G::NTO; ++t) {
            const int tt = t % G::NT;
            const auto& Wsrc = t < G::NT ? Wh : Wv;
            for (int row = 0; row < 16; ++row) {
                const int kout = 16 * tt + row;                    // unit receiving dL/dh (C/D row = natural order)
                if (kout >= H || (tt == NFULL && row >= 4)) continue;
                for (int kq = 0; kq < 4; ++kq) {
                    const int lane = (kq << 4) | row;
                    for (int kk = 0; kk < G::KT; ++kk) {
                        const int u = 4 * kk + kq;
                        if (u >= H) continue;
                        Out::put(&A[(((size_t)t * G::KBG + kk / 2) * 64 + lane) * 2 + (kk & 1)], Wsrc[(size_t)kout * H + u]);
                    }
                }
            }
        }
        double* WD = reinterpret_cast<double*>(img.data() + G::OFF_WD);
        double* BD = reinterpret_cast<double*>(img.data() + G::OFF_BD);
        for (int kt = 0; kt < G::KT; ++kt)
            for (int q = 0; q < 4; ++q) {
                const int unit = 4 * kt + q;
                if (unit >= H) continue;
                Out::put(&WD[(kt * 4 + q) * 2], Wd[(size_t)unit * 2]);
                Out::put(&WD[(kt * 4 + q) * 2 + 1], Wd[(size_t)unit * 2 + 1]);
            }
        Out::put(&BD[0], bd[0]);
        Out::put(&BD[1], bd[1]);
        return img;
    }

    static int run(rnnwf_handle* h, MdGradArgs a, int64_t R, double* dW) {
        const void* fn = (const void*)mdrnn_bwd_kernel<NFULL, WAVES>;
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, G::BYTES, &bpc)) return rc;
        const int64_t need = (a.nsb + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
        if (int rc = ensure(h, h->rowbuf, (size_t)grid * WAVES * 2 * a.Nx * G::KT * 64 * 8)) return rc;
        a.ring = (double*)h->rowbuf.p;
        if (int rc = head_part_alloc<double>(h, (size_t)grid * WAVES, 2 * G::HEAD_ROW, &a.head_part)) return rc;
        {
            TimedLaunch tl(h, 3);
            mdrnn_bwd_kernel<NFULL, WAVES><<<grid, WAVES * 64, G::BYTES, h->stream>>>(a);
            head_reduce_launch<double>(h, (size_t)grid * WAVES, 2 * G::HEAD_ROW, a.head_grad);
        }
        RNNWF_HIP(h, hipGetLastError());
        return tn_gemm_launch<double, G::PCOLS / 16, G::QCOLS / 16>(h, a.P, a.Q, R, dW);
    }

    static void unpack(rnnwf_handle* h, const double* dW, const double* hg) {
        const int H = h->H;
        auto col_of_unit = [&](int k) { return k < 16 * NFULL ? 16 * (k / 16) + 4 * (k % 4) + (k % 16) / 4 : 16 * NFULL + 4 * (k - 16 * NFULL); };
        const int xcol = 16 * NFULL + 1, onecol = 16 * NFULL + 3, voff = 16 * G::NT;
        auto at = [&](int prow, int col) { return dW[(size_t)prow * G::QCOLS + col]; };
        auto& gWh = h->grads["Wh_rnn_0"];
        auto& gUh = h->grads["Uh_rnn_0"];
        auto& gWv = h->grads["Wv_rnn_0"];
        auto& gUv = h->grads["Uv_rnn_0"];
        auto& gb = h->grads["b_rnn_0"];
        auto& gWd = h->grads["wf_dense/kernel"];
        auto& gbd = h->grads["wf_dense/bias"];
        gWh.assign((size_t)H * H, 0.0); gWv.assign((size_t)H * H, 0.0);
        gUh.assign((size_t)2 * H, 0.0); gUv.assign((size_t)2 * H, 0.0);
        gb.assign(H, 0.0); gWd.assign((size_t)H * 2, 0.0); gbd.assign(2, 0.0);
        for (int u = 0; u < H; ++u) {
            const int prow = col_of_unit(u);            // P uses the same tile / quarter / register order as Q
            for (int k = 0; k < H; ++k) {
                gWh[(size_t)k * H + u] = at(prow, col_of_unit(k));
                gWv[(size_t)k * H + u] = at(prow, voff + col_of_unit(k));
            }
            for (int sg = 0; sg < 2; ++sg) {
                gUh[(size_t)sg * H + u] = at(prow, xcol + sg);
                gUv[(size_t)sg * H + u] = at(prow, voff + xcol + sg);
            }
            gb[u] = at(prow, onecol);
            gWd[(size_t)u * 2] = hg[u];
            gWd[(size_t)u * 2 + 1] = hg[G::HEAD_ROW + u];
        }
        gbd[0] = hg[4 * G::KT];
        gbd[1] = hg[G::HEAD_ROW + 4 * G::KT];
    }
};

#define MG_DISPATCH(h, EXPR)                                    \
    do {                                                        \
        switch ((h)->NFULL) {                                   \
            case 1: { using K = MGrad<1, 4>; EXPR; }            \
            case 2: { using K = MGrad<2, 4>; EXPR; }            \
            case 3: { using K = MGrad<3, 4>; EXPR; }            \
            case 4: { using K = MGrad<4, 4>; EXPR; }            \
            case 5: { using K = MGrad<5, 4>; EXPR; }            \
        }                                                       \
    } while (0)

}  // namespace

// The gradient's kernels on the batch of the last rnnwf_vmc_step; result (dW image, then the head rows) left in h->gradW.
// mom_dev != nullptr (device-resident training, train.hip): mean energy and norm come from the step's moments on the device.
int rnnwf::mdrnn_grad_device(rnnwf_handle* h, double mean_energy, double norm, const double* mom_dev, size_t* dw_count) {
    if (h->NFULL > 5) return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_gradient: num_units > 84 not implemented");
    if (h->last_ns <= 0 || !h->last_has_ckpt)
        return h->fail(RNNWF_ERR_STATE, "rnnwf_vmc_gradient: call rnnwf_vmc_step first (its samples, states and E_loc are reused)");
    if (!mom_dev && !(norm > 0)) return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_gradient: norm must be positive");
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    Maps m;
    if (int rc = get_maps(h, &m)) return rc;
    const int N = h->N;
    const int64_t ns = h->last_ns, R = ns * N;
    int pcols = 0, qcols = 0, hgn = 0;
    MG_DISPATCH(h, { pcols = K::G::PCOLS; qcols = K::G::QCOLS; hgn = 2 * K::G::HEAD_ROW; break; });
    if (!h->wbwd_valid) {
        std::vector<char> img;
        MG_DISPATCH(h, { img = K::template pack<double>(h); break; });
        if (int rc = ensure(h, h->wbwd, img.size())) return rc;
        if (int rc = upload(h, h->wbwd.p, img.data(), img.size())) return rc;
        h->wbwd_valid = true;
    }
    if (int rc = ensure(h, h->gradP, (size_t)R * pcols * 8)) return rc;
    if (int rc = ensure(h, h->gradQ, (size_t)R * qcols * 8)) return rc;
    const size_t dwn = (size_t)pcols * qcols + hgn;
    if (dw_count) *dw_count = dwn;
    if (int rc = ensure(h, h->gradW, dwn * 8)) return rc;
    RNNWF_HIP(h, hipMemsetAsync(h->gradW.p, 0, dwn * 8, h->stream));
    MdGradArgs a{};
    a.wbwd = h->wbwd.p;
    a.N = N;
    a.Nx = h->Nx;
    a.ns = ns;
    a.nsb = (ns + kChains - 1) / kChains;
    a.bits = (const uint32_t*)h->bits.p;
    a.hs = (const double*)h->hck.p;
    a.eloc = (const double*)h->eloc.p;
    a.mean_e = mean_energy;
    a.inv_norm = mom_dev ? 1.0 : 1.0 / norm;
    a.mom = mom_dev;
    a.P = (double*)h->gradP.p;
    a.Q = (double*)h->gradQ.p;
    a.head_grad = (double*)h->gradW.p + (size_t)pcols * qcols;
    a.vert_pos = m.vert_pos;
    a.row_first = m.row_first;
    MG_DISPATCH(h, { if (int rc = K::run(h, a, R, (double*)h->gradW.p)) return rc; break; });
    return RNNWF_OK;
}

int rnnwf::mdrnn_vmc_gradient(rnnwf_handle* h, double mean_energy, double norm) {
    size_t dwn = 0;
    if (int rc = mdrnn_grad_device(h, mean_energy, norm, nullptr, &dwn)) return rc;
    int pcols = 0, qcols = 0;
    MG_DISPATCH(h, { pcols = K::G::PCOLS; qcols = K::G::QCOLS; break; });
    if (int rc = ensure_staging(h, dwn * 8)) return rc;
    const double* host = (const double*)h->staging;
    RNNWF_HIP(h, hipMemcpyAsync(h->staging, h->gradW.p, dwn * 8, hipMemcpyDeviceToHost, h->stream));
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    MG_DISPATCH(h, { K::unpack(h, host, host + (size_t)pcols * qcols); break; });
    return RNNWF_OK;
}

// ---- device-resident training (train.hip): the packers' tables and the gradient image's map ---------------------------------
int rnnwf::mdrnn_pack_table(rnnwf_handle* h, bool backward) {
    if (backward) { MG_DISPATCH(h, { K::template pack<Lin>(h); return 0; }); }
    else { MD_DISPATCH(h, { K::template pack<Lin>(h); return 0; }); }
    return 1;
}

// sidx[i] = 1 + the gradient image element parameter i (order of rnnwf_set_params_flat) is read from, found by running the host
// unpacker on an image holding its own indices
int rnnwf::mdrnn_grad_probe(rnnwf_handle* h, std::vector<int32_t>& sidx, size_t* dw_count) {
    int pcols = 0, qcols = 0, hgn = 0;
    MG_DISPATCH(h, { pcols = K::G::PCOLS; qcols = K::G::QCOLS; hgn = 2 * K::G::HEAD_ROW; break; });
    const size_t n = (size_t)pcols * qcols + hgn;
    if (n == 0 || n >= ((size_t)1 << 31)) return h->fail(RNNWF_ERR_INVALID, "gradient image of %zu elements cannot be probed", n);
    const auto saved = h->grads;
    std::vector<double> img(n);
    for (size_t k = 0; k < n; ++k) img[k] = (double)(k + 1);
    MG_DISPATCH(h, { K::unpack(h, img.data(), img.data() + (size_t)pcols * qcols); break; });
    sidx.clear();
    int rc = 0;
    for (auto& kv : h->params) {
        auto it = h->grads.find(kv.first);
        if (it == h->grads.end() || it->second.size() != kv.second.value.size()) {
            rc = h->fail(RNNWF_ERR_STATE, "mdrnn_grad_probe: no gradient for '%s'", kv.first.c_str());
            break;
        }
        for (size_t i = 0; i < kv.second.slot.size(); ++i) sidx.push_back((int32_t)std::llround(it->second[(size_t)kv.second.slot[i]]));
    }
    h->grads = saved;
    if (dw_count) *dw_count = n;
    return rc;
}
