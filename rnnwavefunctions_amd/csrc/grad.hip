// grad.hip - host side of the VMC-cost gradient for the GRU RNNs (f32 1D positive / complex, f64 2D-lattice GRU):
// rnnwf_vmc_gradient / rnnwf_get_grad / rnnwf_allreduce_grads.
#include <algorithm>

#include "grad_kernels.h"
#include "tn_gemm.h"
#include "ml_grad_kernels.h"
#include "models.h"
#include "pack.h"

using namespace rnnwf;

namespace {

template <typename T, int NFULL, int WAVES, int NOUT>
struct GLaunch {
    using L = GruLayout<T, NFULL, NOUT>;
    using G = GradLayout<NFULL, T>;
    static constexpr bool STREAM = GradStream<T, NFULL, NOUT>::value;       // backward operand read through L2 (grad_kernels.h)
    static constexpr size_t LDS = L::LDS_BYTES + (STREAM ? 0 : G::BWD_BYTES);

    template <class S = double>
    static std::vector<char> pack_bwd(const rnnwf_handle* h) {
        using Out = PackSink<S>;
        const int H = h->H;
        std::vector<char> img(G::BWD_BYTES, 0);
        Out::begin(img);
        const std::string pre = kGruPre;
        const auto Wg = pvs<S>(h, pre + "gates/kernel");                         // [2+H, 2H]
        const auto Wch = pvs<S>(h, pre + "candidate/hidden_projection/kernel");  // [H, H]
        T* A = reinterpret_cast<T*>(img.data());
        for (int t = 0; t < G::NTO; ++t)
            for (int row = 0; row < 16; ++row) {
                int q, r;
                row_to_qr<T>(row, q, r);
                if (t == NFULL && r != 0) continue;
                const int kout = t < NFULL ? 16 * t + 4 * r + q : 16 * NFULL + q;   // hidden unit receiving dL/dh
                if (kout >= H) continue;
                for (int kq = 0; kq < 4; ++kq) {
                    const int lane = (kq << 4) | row;
                    for (int kk = 0; kk < G::KB; ++kk) {
                        const int g = kk / G::KT, kt = kk % G::KT, u = 4 * kt + kq;  // pre-activation (gate g, unit u)
                        if (u >= H) continue;
                        S w;
                        if (g == 0) w = Wg[(size_t)(2 + kout) * 2 * H + u];
                        else if (g == 1) w = Wg[(size_t)(2 + kout) * 2 * H + H + u];
                        else w = Wch[(size_t)kout * H + u];
                        Out::put(&A[(((size_t)t * G::KBG + kk / G::VW) * 64 + lane) * G::VW + (kk % G::VW)], w);
                    }
                }
            }
        return img;
    }

    // the wide shapes, whose kernels live in grad_wide.hip's translation unit (see there): -1 otherwise
    static constexpr int WIDE = (std::is_same<T, float>::value && NOUT == 3 && WAVES == 4) ? (NFULL == 8 ? 0 : NFULL == 12 ? 1 : NFULL == 16 ? 3 : -1)
                              : (std::is_same<T, float>::value && NOUT == 1 && WAVES == 4) ? (NFULL == 8 ? 4 : NFULL == 12 ? 5 : NFULL == 16 ? 6 : -1)
                              : (std::is_same<T, double>::value && NOUT == 1 && WAVES == 4 && NFULL == 6) ? 2 : -1;
    static const void* kernel() {
        if constexpr (WIDE >= 0) return grad_wide_kernel(WIDE);
        else return (const void*)gru_bwd_kernel<T, NFULL, WAVES, NOUT>;
    }
    static void launch(unsigned grid, size_t lds, hipStream_t stream, const GradArgs& a) {
        if constexpr (WIDE >= 0) grad_wide_launch(WIDE, grid, lds, stream, a);
        else gru_bwd_kernel<T, NFULL, WAVES, NOUT><<<grid, WAVES * 64, lds, stream>>>(a);
    }

    // small batches: two waves per block of 16 chains (grad_kernels.h: GradPair) while every pair still gets SIMDs of its own
    static constexpr bool PAIR_OK = WIDE < 0 && std::is_same<T, float>::value && WAVES == 4 && GradPair<T, NFULL, NOUT>::FITS;
    static int run_pair(rnnwf_handle* h, GradArgs a, int64_t R, void* dW) {
        if constexpr (PAIR_OK) {
            using GP = GradPair<T, NFULL, NOUT>;
            const void* fn = (const void*)gru_bwd_kernel<T, NFULL, 2 * GP::NB, NOUT, true>;
            int bpc = 0;
            if (int rc = rnnwf::blocks_per_cu(h, fn, 2 * GP::NB * 64, GP::LDS, &bpc)) return rc;
            const int64_t need = (a.nsb + GP::NB - 1) / GP::NB;
            const unsigned grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
            constexpr int HN = NOUT * G::HEAD_ROW;
            T* part = nullptr;
            if (int rc = head_part_alloc<T>(h, (size_t)grid * GP::NB, HN, &part)) return rc;
            a.head_part = part;
            {
                TimedLaunch tl(h, 3);
                gru_bwd_kernel<T, NFULL, 2 * GP::NB, NOUT, true><<<grid, 2 * GP::NB * 64, GP::LDS, h->stream>>>(a);
                head_reduce_launch<T>(h, (size_t)grid * GP::NB, HN, (T*)a.head_grad);
            }
            RNNWF_HIP(h, hipGetLastError());
            return tn_gemm_launch<T, G::PCOLS / 16, G::QCOLS / 16>(h, (const T*)a.P, (const T*)a.Q, R, (T*)dW);
        }
        return RNNWF_ERR_INVALID;
    }

    static int run(rnnwf_handle* h, GradArgs a, int64_t R, void* dW) {
        if constexpr (PAIR_OK)
            if (a.nsb <= (int64_t)GradPair<T, NFULL, NOUT>::NB * h->cu_count && !h->knobs.no_coop) return run_pair(h, a, R, dW);
        const void* fn = kernel();
        const size_t lds = LDS;
        if (lds > 160 * 1024)
            return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_gradient: forward + backward weight images (%zu B) exceed the 160 KB LDS", lds);
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, lds, &bpc)) return rc;
        const int64_t need = (a.nsb + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
        constexpr int HN = NOUT * G::HEAD_ROW;
        T* part = nullptr;
        if (int rc = head_part_alloc<T>(h, (size_t)grid * WAVES, HN, &part)) return rc;
        a.head_part = part;
        {
            TimedLaunch tl(h, 3);
            launch(grid, lds, h->stream, a);
            head_reduce_launch<T>(h, (size_t)grid * WAVES, HN, (T*)a.head_grad);
        }
        RNNWF_HIP(h, hipGetLastError());
        return tn_gemm_launch<T, G::PCOLS / 16, G::QCOLS / 16>(h, (const T*)a.P, (const T*)a.Q, R, (T*)dW);
    }

    // dW image [PCOLS][QCOLS] + head gradients -> TF-named gradient arrays
    static void unpack(rnnwf_handle* h, const void* dW_, size_t head_off) {
        const T* dW = (const T*)dW_;
        const T* hg = dW + head_off;
        const int H = h->H;
        const int NT = G::NT;
        auto col_of_unit = [&](int k) { return k < 16 * NFULL ? 16 * (k / 16) + 4 * (k % 4) + (k % 16) / 4 : 16 * NFULL + 4 * (k - 16 * NFULL); };
        const int xcol = 16 * NFULL + 1, onecol = 16 * NFULL + 3;
        auto at = [&](int prow, int col) { return (double)dW[(size_t)prow * G::QCOLS + col]; };
        const std::string pre = kGruPre;
        auto& gWg = h->grads[pre + "gates/kernel"];
        auto& gbg = h->grads[pre + "gates/bias"];
        auto& gWci = h->grads[pre + "candidate/input_projection/kernel"];
        auto& gbci = h->grads[pre + "candidate/input_projection/bias"];
        auto& gWch = h->grads[pre + "candidate/hidden_projection/kernel"];
        auto& gbch = h->grads[pre + "candidate/hidden_projection/bias"];
        const char* amp = NOUT == 1 ? "wf_dense" : "wf_dense_ampl";
        auto& gWd = h->grads[std::string(amp) + "/kernel"];
        auto& gbd = h->grads[std::string(amp) + "/bias"];
        gWg.assign((size_t)(2 + H) * 2 * H, 0.0); gbg.assign(2 * H, 0.0);
        gWci.assign((size_t)2 * H, 0.0); gbci.assign(H, 0.0);
        gWch.assign((size_t)H * H, 0.0); gbch.assign(H, 0.0);
        gWd.assign((size_t)H * 2, 0.0); gbd.assign(2, 0.0);
        std::vector<double>* gWp = nullptr;
        std::vector<double>* gbp = nullptr;
        if (NOUT == 3) {
            gWp = &h->grads["wf_dense_phase/kernel"];
            gbp = &h->grads["wf_dense_phase/bias"];
            gWp->assign((size_t)H * 2, 0.0);
            gbp->assign(2, 0.0);
        }
        for (int u = 0; u < H; ++u) {
            const int m = u / 16, r = (u % 16) / 4, q = u % 4;
            const bool full = u < 16 * NFULL;
            const int uq = u - 16 * NFULL;                       // remainder units: lane quarter = uq
            int prow[3];
            for (int g = 0; g < 3; ++g) prow[g] = full ? (g * NFULL + m) * 16 + 4 * q + r : (NT - 1) * 16 + 4 * uq + g;
            const int prow_y = full ? (NT + m) * 16 + 4 * q + r : (NT + NFULL) * 16 + 4 * uq;
            for (int k = 0; k < H; ++k) {
                const int col = col_of_unit(k);
                gWg[(size_t)(2 + k) * 2 * H + u] = at(prow[0], col);
                gWg[(size_t)(2 + k) * 2 * H + H + u] = at(prow[1], col);
                gWch[(size_t)k * H + u] = at(prow[2], col);
            }
            for (int sgm = 0; sgm < 2; ++sgm) {
                gWg[(size_t)sgm * 2 * H + u] = at(prow[0], xcol + sgm);
                gWg[(size_t)sgm * 2 * H + H + u] = at(prow[1], xcol + sgm);
                gWci[(size_t)sgm * H + u] = at(prow_y, xcol + sgm);
            }
            gbg[u] = at(prow[0], onecol);
            gbg[H + u] = at(prow[1], onecol);
            gbch[u] = at(prow[2], onecol);
            gbci[u] = at(prow_y, onecol);
            // head: the image holds the logit difference z1 - z0, so d/dWd[:,1] = +v and d/dWd[:,0] = -v
            const double v = hg[u];          // slot 4 kt + q == unit index
            gWd[(size_t)u * 2 + 1] = v;
            gWd[(size_t)u * 2] = -v;
            if (NOUT == 3) {
                (*gWp)[(size_t)u * 2] = hg[G::HEAD_ROW + u];
                (*gWp)[(size_t)u * 2 + 1] = hg[2 * G::HEAD_ROW + u];
            }
        }
        gbd[1] = hg[4 * G::KT];
        gbd[0] = -hg[4 * G::KT];
        if (NOUT == 3) {
            (*gbp)[0] = hg[G::HEAD_ROW + 4 * G::KT];
            (*gbp)[1] = hg[2 * G::HEAD_ROW + 4 * G::KT];
        }
    }
};

#define GRAD_DISPATCH(h, EXPR)                                          \
    do {                                                                \
        if ((h)->model == RNNWF_MODEL_CRNN_U1) {                        \
            switch ((h)->NFULL) {                                       \
                case 1: { using K = GLaunch<float, 1, 4, 3>; EXPR; }    \
                case 2: { using K = GLaunch<float, 2, 4, 3>; EXPR; }    \
                case 3: { using K = GLaunch<float, 3, 4, 3>; EXPR; }    \
                case 4: { using K = GLaunch<float, 4, 4, 3>; EXPR; }    \
                case 6: { using K = GLaunch<float, 6, 4, 3>; EXPR; }    \
                case 8: { using K = GLaunch<float, 8, 4, 3>; EXPR; }    /* these two and the float64 GRU's widest: the      */ \
                case 12: { using K = GLaunch<float, 12, 4, 3>; EXPR; }  /* kernels live in grad_wide.hip (GLaunch::WIDE)    */ \
                case 16: { using K = GLaunch<float, 16, 4, 3>; EXPR; }  \
            }                                                           \
        } else if ((h)->model == RNNWF_MODEL_GRU1D_F64) {               \
            switch ((h)->NFULL) {                                       \
                case 1: { using K = GLaunch<double, 1, 4, 1>; EXPR; }   \
                case 2: { using K = GLaunch<double, 2, 4, 1>; EXPR; }   \
                case 3: { using K = GLaunch<double, 3, 4, 1>; EXPR; }   \
                case 4: { using K = GLaunch<double, 4, 4, 1>; EXPR; }   \
                case 6: { using K = GLaunch<double, 6, 4, 1>; EXPR; }   /* kernel in grad_wide.hip */ \
            }                                                           \
        } else {                                                        \
            switch ((h)->NFULL) {                                       \
                case 1: { using K = GLaunch<float, 1, 4, 1>; EXPR; }    \
                case 2: { using K = GLaunch<float, 2, 4, 1>; EXPR; }    \
                case 3: { using K = GLaunch<float, 3, 4, 1>; EXPR; }    \
                case 4: { using K = GLaunch<float, 4, 4, 1>; EXPR; }    \
                case 6: { using K = GLaunch<float, 6, 4, 1>; EXPR; }    \
                case 8: { using K = GLaunch<float, 8, 4, 1>; EXPR; }    \
                case 12: { using K = GLaunch<float, 12, 4, 1>; EXPR; }  \
                case 16: { using K = GLaunch<float, 16, 4, 1>; EXPR; }  \
            }                                                           \
        }                                                               \
    } while (0)

// ---- stacked layers: one backward pass per layer, top first (ml_grad_kernels.h) ---------------------------------
// NOUT = 1: positive RNN; NOUT = 3: complex RNN (heads on the top layer, complex weights w_s as in gru_bwd_kernel).
// T = float, or double for the 2D-lattice GRU (NOUT = 1).
template <int NFULL, int NL, int WAVES, int NOUT = 1, typename T = float>
struct MLGrad {
    using G0 = GLaunch<T, NFULL, WAVES, NOUT>;
    using L0 = GruLayout<T, NFULL, NOUT>;
    using U = UpperLayout<NFULL, T>;
    using GU = UpperGradLayout<NFULL, NOUT, T>;
    static constexpr size_t ES = sizeof(T);
    static constexpr size_t DW0 = (size_t)G0::G::PCOLS * G0::G::QCOLS;     // elements
    static constexpr size_t HEAD = (size_t)NOUT * G0::G::HEAD_ROW;
    static constexpr size_t DWU = (size_t)GU::PCOLS * GU::QCOLS;
    static constexpr size_t DW_FLOATS = DW0 + HEAD + (NL - 1) * DWU;       // [dW layer 0 | head | dW layer 1 | ...]

    template <class S = double>
    static std::vector<char> pack_upper_bwd(const rnnwf_handle* h, int layer) {
        using Out = PackSink<S>;
        const int H = h->H;
        std::vector<char> img(GU::BWD_BYTES, 0);
        Out::begin(img);
        const std::string pre = "multi_rnn_cell/cell_" + std::to_string(layer) + "/cudnn_compatible_gru_cell/";
        const auto Wg = pvs<S>(h, pre + "gates/kernel");                         // [H + H, 2H]
        const auto Wci = pvs<S>(h, pre + "candidate/input_projection/kernel");   // [H, H]
        const auto Wch = pvs<S>(h, pre + "candidate/hidden_projection/kernel");  // [H, H]
        for (int side = 0; side < 2; ++side) {                                // 0: H side (-> dh), 1: X side (-> dx)
            T* A = reinterpret_cast<T*>(img.data() + side * GU::SIDE_BYTES);
            for (int t = 0; t < GU::NTO; ++t)
                for (int row = 0; row < 16; ++row) {
                    int q, r;
                    row_to_qr<T>(row, q, r);
                    if (t == NFULL && r != 0) continue;
                    const int kout = t < NFULL ? 16 * t + 4 * r + q : 16 * NFULL + q;
                    if (kout >= H) continue;
                    const size_t grow = side == 0 ? (size_t)H + kout : (size_t)kout;   // row of the gates kernel
                    for (int kq = 0; kq < 4; ++kq) {
                        const int lane = (kq << 4) | row;
                        for (int kk = 0; kk < GU::KB; ++kk) {
                            const int g = kk / GU::KT, kt = kk % GU::KT, u = 4 * kt + kq;
                            if (u >= H) continue;
                            S w;
                            if (g == 0) w = Wg[grow * 2 * H + u];
                            else if (g == 1) w = Wg[grow * 2 * H + H + u];
                            else if (side == 0) w = Wch[(size_t)kout * H + u];
                            else w = Wci[(size_t)kout * H + u];
                            Out::put(&A[(((size_t)t * GU::KBG + kk / GU::VW) * 64 + lane) * GU::VW + (kk % GU::VW)], w);
                        }
                    }
                }
        }
        return img;
    }

    static std::vector<char> pack_all(const rnnwf_handle* h) {
        std::vector<char> img = G0::template pack_bwd<double>(h);
        for (int l = 1; l < NL; ++l) {
            const std::vector<char> up = pack_upper_bwd<double>(h, l);
            img.insert(img.end(), up.begin(), up.end());
        }
        return img;
    }
    // the same images once more over Lin: the table of the whole backward buffer (pack_value.h; the active PackTrace's shift moves along)
    static void pack_all_table(const rnnwf_handle* h) {
        const size_t base = pack_trace().shift;
        G0::template pack_bwd<Lin>(h);
        for (int l = 1; l < NL; ++l) {
            pack_trace().shift = base + G0::G::BWD_BYTES + (size_t)(l - 1) * GU::BWD_BYTES;
            pack_upper_bwd<Lin>(h, l);
        }
        pack_trace().shift = base;
    }
    // and of the forward buffer [layer 0 | upper layers]
    static void pack_forward_table(const rnnwf_handle* h) {
        const size_t base = pack_trace().shift;
        pack_gru_image<T, NFULL, NOUT, Lin>(h);
        for (int l = 1; l < NL; ++l) {
            pack_trace().shift = base + L0::BYTES + (size_t)(l - 1) * U::BYTES;
            pack_upper_image<NFULL, T, Lin>(h, l);
        }
        pack_trace().shift = base;
    }
    static void probe_unpack(rnnwf_handle* h, size_t* count) {
        std::vector<T> img(DW_FLOATS);
        for (size_t k = 0; k < DW_FLOATS; ++k) img[k] = (T)(k + 1);
        G0::unpack(h, img.data(), DW0);
        for (int l = 1; l < NL; ++l) unpack_upper(h, img.data() + DW0 + HEAD + (size_t)(l - 1) * DWU, l);
        *count = DW_FLOATS;
    }

    template <bool TOP>
    static int upper_pass(rnnwf_handle* h, UpperGradArgs a) {
        const void* fn = (const void*)gru_upper_bwd_kernel<T, NFULL, WAVES, TOP, NOUT>;
        const size_t lds = GU::WIDE ? GU::HEAD_BYTES : U::BYTES + GU::BWD_BYTES + (TOP ? GU::HEAD_BYTES : 0);
        if (lds > 160 * 1024) return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_gradient: stacked-layer images (%zu B) exceed the 160 KB LDS", lds);
        int bpc = 0;
        if (int rc = rnnwf::blocks_per_cu(h, fn, WAVES * 64, lds, &bpc)) return rc;
        const int64_t need = (a.nsb + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)std::min<int64_t>(need, (int64_t)bpc * h->cu_count);
        constexpr int HN = NOUT * GU::HEAD_ROW;
        if (TOP) {
            T* part = nullptr;
            if (int rc = head_part_alloc<T>(h, (size_t)grid * WAVES, HN, &part)) return rc;
            a.head_part = part;
        }
        {
            TimedLaunch tl(h, 3);
            gru_upper_bwd_kernel<T, NFULL, WAVES, TOP, NOUT><<<grid, WAVES * 64, lds, h->stream>>>(a);
            if (TOP) head_reduce_launch<T>(h, (size_t)grid * WAVES, HN, (T*)a.head_grad);
        }
        RNNWF_HIP(h, hipGetLastError());
        return 0;
    }

    static int gemm(rnnwf_handle* h, const T* P, const T* Q, int64_t R, T* dW, bool upper) {
        if (upper) return tn_gemm_launch<T, GU::PCOLS / 16, GU::QCOLS / 16>(h, P, Q, R, dW);
        return tn_gemm_launch<T, G0::G::PCOLS / 16, G0::G::QCOLS / 16>(h, P, Q, R, dW);
    }

    static int run(rnnwf_handle* h, double mean_energy, double mean_energy_im, double norm) {
        if (int rc = run_device(h, mean_energy, mean_energy_im, norm, nullptr)) return rc;
        if (int rc = ensure_staging(h, DW_FLOATS * ES)) return rc;
        const T* host = (const T*)h->staging;
        RNNWF_HIP(h, hipMemcpyAsync(h->staging, h->gradW.p, DW_FLOATS * ES, hipMemcpyDeviceToHost, h->stream));
        RNNWF_HIP(h, hipStreamSynchronize(h->stream));
        G0::unpack(h, host, DW0);                              // layer 0 + head (written by the top layer's pass)
        for (int l = 1; l < NL; ++l) unpack_upper(h, host + DW0 + HEAD + (size_t)(l - 1) * DWU, l);
        return RNNWF_OK;
    }

    // every kernel of the stacked gradient, result left in h->gradW; mom_dev != nullptr: device-resident training (grad_single_layer_device)
    static int run_device(rnnwf_handle* h, double mean_energy, double mean_energy_im, double norm, const double* mom_dev) {
        const int N = h->N;
        const int64_t ns = h->last_ns, R = ns * N, nsb = (ns + kChains - 1) / kChains;
        const double inv_norm = mom_dev ? (NOUT == 3 ? 2.0 : 1.0) : (NOUT == 3 ? 2.0 : 1.0) / norm;     // the complex cost carries a factor 2 (TrainingRNN_J1J2.py:197)
        const bool parity = h->model == RNNWF_MODEL_GRU1D_PARITY;
        if (!h->wbwd_valid) {
            const std::vector<char> img = pack_all(h);
            if (int rc = ensure(h, h->wbwd, img.size())) return rc;
            if (int rc = upload(h, h->wbwd.p, img.data(), img.size())) return rc;
            h->wbwd_valid = true;
        }
        if (int rc = ensure(h, h->gradP, (size_t)R * GU::PCOLS * ES)) return rc;
        if (int rc = ensure(h, h->gradQ, (size_t)R * GU::QCOLS * ES)) return rc;
        if (int rc = ensure(h, h->gradW, (DW_FLOATS + HEAD) * ES)) return rc;     // + a scratch head row for layer 0's pass
        const size_t dx_bytes = (size_t)N * nsb * L0::KT * 64 * ES;
        for (int i = 0; i < (NL > 2 ? 2 : 1); ++i) if (int rc = ensure(h, h->gradDX[i], dx_bytes)) return rc;
        RNNWF_HIP(h, hipMemsetAsync(h->gradW.p, 0, (DW_FLOATS + HEAD) * ES, h->stream));
        if (parity) {
            // log P_sym = log(0.5 (P_F + P_R)): both directions, teacher-forced, each sample weighted by the direction's share of
            // P_sym (the single-layer case in rnnwf_vmc_gradient below explains the sequence)
            if constexpr (NOUT == 1 && sizeof(T) == 4) {
                if (int rc = ensure(h, h->out_lp, (size_t)ns * 8)) return rc;
                if (int rc = ensure(h, h->out_lp2, (size_t)ns * 8)) return rc;
                double* lpF = (double*)h->out_lp.p;
                double* lpR = (double*)h->out_lp2.p;
                if (int rc = prnn_teacher_base(h, ns, false, lpF)) return rc;
                if (int rc = prnn_teacher_base(h, ns, true, lpR)) return rc;
                if (int rc = run_parity_share(h, lpF, lpR, ns)) return rc;
                if (int rc = passes(h, mean_energy, mean_energy_im, inv_norm, (const uint32_t*)h->bits2.p, lpR, mom_dev)) return rc;
                if (int rc = prnn_teacher_base(h, ns, false, nullptr)) return rc;
                if (int rc = passes(h, mean_energy, mean_energy_im, inv_norm, (const uint32_t*)h->bits.p, lpF, mom_dev)) return rc;
            }
        } else {
            if (int rc = passes(h, mean_energy, mean_energy_im, inv_norm, (const uint32_t*)h->bits.p, nullptr, mom_dev)) return rc;
        }
        return RNNWF_OK;
    }

    // one backward pass per layer, top first, over the resident checkpoints of the chains in `bits`; everything is ADDED to gradW
    static int passes(rnnwf_handle* h, double mean_energy, double mean_energy_im, double inv_norm, const uint32_t* bits, const double* wfac,
                      const double* mom_dev) {
        const int N = h->N;
        const int64_t ns = h->last_ns, R = ns * N, nsb = (ns + kChains - 1) / kChains;
        T* dW = (T*)h->gradW.p;
        const char* bwd = (const char*)h->wbwd.p;
        const void* dh_in = nullptr;
        for (int l = NL - 1; l >= 1; --l) {
            UpperGradArgs a{};
            a.wup = (const char*)h->wimg.p + L0::BYTES + (size_t)(l - 1) * U::BYTES;
            a.wbwd = bwd + G0::G::BWD_BYTES + (size_t)(l - 1) * GU::BWD_BYTES;
            a.whead = (const char*)h->wimg.p + L0::OFF_WD;
            a.N = N; a.layer = l; a.hck_nl = NL;
            a.ns = ns; a.nsb = nsb;
            a.bits = bits;
            a.wfac = wfac;
            a.hck = h->hck.p;
            a.eloc = (const double*)h->eloc.p;
            a.eloc_c = (const float2*)h->eloc.p;
            a.mean_e = mean_energy;
            a.mean_im = mean_energy_im;
            a.inv_norm = inv_norm;
            a.mom = mom_dev;
            a.dh_in = dh_in;
            a.dx_out = h->gradDX[(NL - 1 - l) & 1].p;
            a.P = h->gradP.p;
            a.Q = h->gradQ.p;
            a.head_grad = dW + DW0;
            if (l == NL - 1) { if (int rc = upper_pass<true>(h, a)) return rc; }
            else { if (int rc = upper_pass<false>(h, a)) return rc; }
            if (int rc = gemm(h, (const T*)a.P, (const T*)a.Q, R, dW + DW0 + HEAD + (size_t)(l - 1) * DWU, true)) return rc;
            dh_in = a.dx_out;
        }
        GradArgs a{};
        a.wimg = h->wimg.p;
        a.wbwd = h->wbwd.p;
        a.N = N; a.ns = ns; a.nsb = nsb;
        a.bits = bits;
        a.hck = h->hck.p;
        a.eloc = (const double*)h->eloc.p;
        a.eloc_c = (const float2*)h->eloc.p;
        a.mean_e = mean_energy;
        a.mean_im = mean_energy_im;
        a.inv_norm = inv_norm;
        a.mom = mom_dev;
        a.P = h->gradP.p;
        a.Q = h->gradQ.p;
        a.head_grad = dW + DW_FLOATS;          // scratch rows: layer 0 has no head term here (its adds are zeros)
        a.dh_in = dh_in;
        a.hck_nl = NL;
        return G0::run(h, a, R, dW);
    }

    static void unpack_upper(rnnwf_handle* h, const T* dW, int layer) {
        const int H = h->H;
        const int NT = GU::NT;
        auto col_of_unit = [&](int k) { return k < 16 * NFULL ? 16 * (k / 16) + 4 * (k % 4) + (k % 16) / 4 : 16 * NFULL + 4 * (k - 16 * NFULL); };
        const int hoff = 16 * GU::NTO, onecol = hoff + 16 * NFULL + 1;
        auto at = [&](int prow, int col) { return (double)dW[(size_t)prow * GU::QCOLS + col]; };
        const std::string pre = "multi_rnn_cell/cell_" + std::to_string(layer) + "/cudnn_compatible_gru_cell/";
        auto& gWg = h->grads[pre + "gates/kernel"];
        auto& gbg = h->grads[pre + "gates/bias"];
        auto& gWci = h->grads[pre + "candidate/input_projection/kernel"];
        auto& gbci = h->grads[pre + "candidate/input_projection/bias"];
        auto& gWch = h->grads[pre + "candidate/hidden_projection/kernel"];
        auto& gbch = h->grads[pre + "candidate/hidden_projection/bias"];
        gWg.assign((size_t)2 * H * 2 * H, 0.0); gbg.assign(2 * H, 0.0);
        gWci.assign((size_t)H * H, 0.0); gbci.assign(H, 0.0);
        gWch.assign((size_t)H * H, 0.0); gbch.assign(H, 0.0);
        for (int u = 0; u < H; ++u) {
            const int m = u / 16, r = (u % 16) / 4, q = u % 4;
            const bool full = u < 16 * NFULL;
            const int uq = u - 16 * NFULL;
            int prow[3];
            for (int g = 0; g < 3; ++g) prow[g] = full ? (g * NFULL + m) * 16 + 4 * q + r : (NT - 1) * 16 + 4 * uq + g;
            const int prow_y = full ? (NT + m) * 16 + 4 * q + r : (NT + NFULL) * 16 + 4 * uq;
            for (int k = 0; k < H; ++k) {
                const int cx = col_of_unit(k), ch = hoff + col_of_unit(k);
                gWg[(size_t)k * 2 * H + u] = at(prow[0], cx);
                gWg[(size_t)k * 2 * H + H + u] = at(prow[1], cx);
                gWg[(size_t)(H + k) * 2 * H + u] = at(prow[0], ch);
                gWg[(size_t)(H + k) * 2 * H + H + u] = at(prow[1], ch);
                gWci[(size_t)k * H + u] = at(prow_y, cx);
                gWch[(size_t)k * H + u] = at(prow[2], ch);
            }
            gbg[u] = at(prow[0], onecol);
            gbg[H + u] = at(prow[1], onecol);
            gbch[u] = at(prow[2], onecol);
            gbci[u] = at(prow_y, onecol);
        }
    }
};

#define MLGRAD_DISPATCH_(h, NOUT, EXPR)                                 \
    do {                                                                \
        if ((h)->NL == 2) {                                             \
            switch ((h)->NFULL) {                                       \
                case 1: { using K = MLGrad<1, 2, 4, NOUT>; EXPR; }      \
                case 2: { using K = MLGrad<2, 2, 4, NOUT>; EXPR; }      \
                case 3: { using K = MLGrad<3, 2, 4, NOUT>; EXPR; }      \
                case 4: { using K = MLGrad<4, 2, 4, NOUT>; EXPR; }      \
                case 6: { using K = MLGrad<6, 2, 4, NOUT>; EXPR; }      \
            }                                                           \
        } else if ((h)->NL == 4) {                                      \
            switch ((h)->NFULL) {                                       \
                case 1: { using K = MLGrad<1, 4, 4, NOUT>; EXPR; }      \
                case 2: { using K = MLGrad<2, 4, 4, NOUT>; EXPR; }      \
                case 3: { using K = MLGrad<3, 4, 4, NOUT>; EXPR; }      \
                case 4: { using K = MLGrad<4, 4, 4, NOUT>; EXPR; }      \
                case 6: { using K = MLGrad<6, 4, 4, NOUT>; EXPR; }      \
            }                                                           \
        } else if ((h)->NL == 3) {                                      \
            switch ((h)->NFULL) {                                       \
                case 1: { using K = MLGrad<1, 3, 4, NOUT>; EXPR; }      \
                case 2: { using K = MLGrad<2, 3, 4, NOUT>; EXPR; }      \
                case 3: { using K = MLGrad<3, 3, 4, NOUT>; EXPR; }      \
                case 4: { using K = MLGrad<4, 3, 4, NOUT>; EXPR; }      \
                case 6: { using K = MLGrad<6, 3, 4, NOUT>; EXPR; }      \
            }                                                           \
        }                                                               \
    } while (0)
#define MLGRAD_DISPATCH(h, EXPR)                                        \
    do {                                                                \
        if ((h)->model == RNNWF_MODEL_CRNN_U1) MLGRAD_DISPATCH_(h, 3, EXPR); \
        else if ((h)->model == RNNWF_MODEL_GRU1D_F64) {                 \
            if ((h)->NL == 2 && (h)->NFULL == 1) { using K = MLGrad<1, 2, 4, 1, double>; EXPR; } \
            if ((h)->NL == 2 && (h)->NFULL == 2) { using K = MLGrad<2, 2, 4, 1, double>; EXPR; } \
            if ((h)->NL == 3 && (h)->NFULL == 1) { using K = MLGrad<1, 3, 4, 1, double>; EXPR; } \
            if ((h)->NL == 3 && (h)->NFULL == 2) { using K = MLGrad<2, 3, 4, 1, double>; EXPR; } \
            if ((h)->NL == 2 && (h)->NFULL == 3) { using K = MLGrad<3, 2, 4, 1, double>; EXPR; } \
            if ((h)->NL == 2 && (h)->NFULL == 4) { using K = MLGrad<4, 2, 4, 1, double>; EXPR; } \
            if ((h)->NL == 3 && (h)->NFULL == 3) { using K = MLGrad<3, 3, 4, 1, double>; EXPR; } \
            if ((h)->NL == 3 && (h)->NFULL == 4) { using K = MLGrad<4, 3, 4, 1, double>; EXPR; } \
            if ((h)->NL == 4 && (h)->NFULL == 1) { using K = MLGrad<1, 4, 4, 1, double>; EXPR; } \
            if ((h)->NL == 4 && (h)->NFULL == 2) { using K = MLGrad<2, 4, 4, 1, double>; EXPR; } \
            if ((h)->NL == 4 && (h)->NFULL == 3) { using K = MLGrad<3, 4, 4, 1, double>; EXPR; } \
            if ((h)->NL == 4 && (h)->NFULL == 4) { using K = MLGrad<4, 4, 4, 1, double>; EXPR; } \
        } else MLGRAD_DISPATCH_(h, 1, EXPR);                            \
    } while (0)

}  // namespace

extern "C" int rnnwf_vmc_gradient(rnnwf_handle* h, double mean_energy, double mean_energy_im, double norm) {
    if (!h) return RNNWF_ERR_INVALID;
    if (!h->committed) return h->fail(RNNWF_ERR_STATE, "parameters not committed");
    if (h->model == RNNWF_MODEL_MDRNN2D) return mdrnn_vmc_gradient(h, mean_energy, norm);
    if (h->NL != 1) {
        if (h->last_ns <= 0 || !h->last_has_ckpt)
            return h->fail(RNNWF_ERR_STATE, "rnnwf_vmc_gradient: call rnnwf_vmc_step first (its samples, states and E_loc are reused)");
        if (!(norm > 0)) return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_gradient: norm must be positive");
        RNNWF_HIP(h, hipSetDevice(h->cfg.device));
        MLGRAD_DISPATCH(h, return K::run(h, mean_energy, mean_energy_im, norm));
        return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_gradient: no stacked-layer kernel for this width");
    }
    size_t dw_floats = 0;
    if (int rc = grad_single_layer_device(h, mean_energy, mean_energy_im, norm, nullptr, &dw_floats)) return rc;
    const size_t es = h->model == RNNWF_MODEL_GRU1D_F64 ? 8 : 4;
    int pcols = 0, qcols = 0;
    GRAD_DISPATCH(h, { pcols = K::G::PCOLS; qcols = K::G::QCOLS; break; });
    if (int rc = ensure_staging(h, dw_floats * es)) return rc;        // pinned: the copy is a plain DMA, the one wait is ours
    RNNWF_HIP(h, hipMemcpyAsync(h->staging, h->gradW.p, dw_floats * es, hipMemcpyDeviceToHost, h->stream));
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    GRAD_DISPATCH(h, { K::unpack(h, h->staging, (size_t)pcols * qcols); break; });
    return RNNWF_OK;
}

// The single-layer gradient's kernels on the batch of the last rnnwf_vmc_step: back-propagation through time + weight-gradient GEMM,
// result (the dW image and the head rows) left in h->gradW.  mom_dev != nullptr (device-resident training, train.hip): mean energy
// and norm come from the step's moments on the device (mean_energy / norm arguments unused) and nothing visits the host.
int rnnwf::grad_single_layer_device(rnnwf_handle* h, double mean_energy, double mean_energy_im, double norm, const double* mom_dev,
                                    size_t* dw_count) {
    if (h->NL != 1) {                       // stacked layers: one backward pass per layer, top first (MLGrad)
        if (h->last_ns <= 0 || !h->last_has_ckpt)
            return h->fail(RNNWF_ERR_STATE, "rnnwf_vmc_gradient: call rnnwf_vmc_step first (its samples, states and E_loc are reused)");
        RNNWF_HIP(h, hipSetDevice(h->cfg.device));
        MLGRAD_DISPATCH(h, { if (dw_count) *dw_count = K::DW_FLOATS; return K::run_device(h, mean_energy, mean_energy_im, norm, mom_dev); });
        return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_gradient: no stacked-layer kernel for this width");
    }
    const bool parity = h->model == RNNWF_MODEL_GRU1D_PARITY;
    const bool cplx = h->model == RNNWF_MODEL_CRNN_U1;
    const bool f64 = h->model == RNNWF_MODEL_GRU1D_F64;
    const size_t es = f64 ? 8 : 4;
    if (h->last_ns <= 0 || !h->last_has_ckpt)
        return h->fail(RNNWF_ERR_STATE, "rnnwf_vmc_gradient: call rnnwf_vmc_step first (its samples, states and E_loc are reused)");
    if (!mom_dev && !(norm > 0)) return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_gradient: norm must be positive");
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    const int N = h->N;
    const int64_t ns = h->last_ns, R = ns * N;
    int pcols = 0, qcols = 0, hgn = 0;
    GRAD_DISPATCH(h, { pcols = K::G::PCOLS; qcols = K::G::QCOLS; hgn = K::G::HEAD_ROW * (cplx ? 3 : 1); break; });
    if (!h->wbwd_valid) {
        std::vector<char> img;
        GRAD_DISPATCH(h, { img = K::template pack_bwd<double>(h); break; });
        if (int rc = ensure(h, h->wbwd, img.size())) return rc;
        if (int rc = upload(h, h->wbwd.p, img.data(), img.size())) return rc;
        h->wbwd_valid = true;
    }
    if (int rc = ensure(h, h->gradP, (size_t)R * pcols * es)) return rc;
    if (int rc = ensure(h, h->gradQ, (size_t)R * qcols * es)) return rc;
    const size_t dw_floats = (size_t)pcols * qcols + hgn;
    if (dw_count) *dw_count = dw_floats;
    if (int rc = ensure(h, h->gradW, dw_floats * es)) return rc;
    RNNWF_HIP(h, hipMemsetAsync(h->gradW.p, 0, dw_floats * es, h->stream));
    GradArgs a{};
    a.wimg = h->wimg.p;
    a.wbwd = h->wbwd.p;
    a.N = N;
    a.ns = ns;
    a.nsb = (ns + kChains - 1) / kChains;
    a.bits = (const uint32_t*)h->bits.p;
    a.hck = h->hck.p;
    a.eloc = (const double*)h->eloc.p;
    a.eloc_c = (const float2*)h->eloc.p;
    a.mean_e = mean_energy;
    a.mean_im = mean_energy_im;
    a.mom = mom_dev;
    a.inv_norm = mom_dev ? (cplx ? 2.0 : 1.0) : (cplx ? 2.0 : 1.0) / norm;      // the complex cost carries a factor 2 (TrainingRNN_J1J2.py:197)
    a.P = h->gradP.p;
    a.Q = h->gradQ.p;
    a.head_grad = (char*)h->gradW.p + (size_t)pcols * qcols * es;
    if (parity) {
        // log P_sym = log(0.5 (P_F + P_R)) (1DTFIM/RNNwavefunction_paritysym.py:145): the gradient is the sum of the two directions'
        // gradients, each sample weighted by the direction's share of P_sym.  The step left the checkpoints of ONE direction: both
        // are redone here, teacher-forced, with the shares from the same two passes.
        if (int rc = ensure(h, h->out_lp, (size_t)ns * 8)) return rc;
        if (int rc = ensure(h, h->out_lp2, (size_t)ns * 8)) return rc;
        double* lpF = (double*)h->out_lp.p;
        double* lpR = (double*)h->out_lp2.p;
        if (int rc = prnn_teacher_base(h, ns, false, lpF)) return rc;
        if (int rc = prnn_teacher_base(h, ns, true, lpR)) return rc;              // the reversed chains' states are resident now
        if (int rc = run_parity_share(h, lpF, lpR, ns)) return rc;
        a.bits = (const uint32_t*)h->bits2.p;
        a.wfac = lpR;
        GRAD_DISPATCH(h, { if (int rc = K::run(h, a, R, h->gradW.p)) return rc; break; });
        if (int rc = prnn_teacher_base(h, ns, false, nullptr)) return rc;
        a.bits = (const uint32_t*)h->bits.p;
        a.wfac = lpF;
    }
    GRAD_DISPATCH(h, { if (int rc = K::run(h, a, R, h->gradW.p)) return rc; break; });
    return RNNWF_OK;
}

// ---- what train.hip needs from this translation unit (the layouts live in its anonymous namespace) -------------------------
// table of a stack's forward buffer [layer 0 | upper layers] into the active PackTrace
int rnnwf::grad_stack_forward_table(rnnwf_handle* h) {
    MLGRAD_DISPATCH(h, { K::pack_forward_table(h); return 0; });
    return h->fail(RNNWF_ERR_INVALID, "no stacked-layer layout for NFULL=%d", h->NFULL);
}
// table of the backward image (pack_value.h) into the active PackTrace
int rnnwf::grad_bwd_pack_table(rnnwf_handle* h) {
    if (h->NL != 1) {
        MLGRAD_DISPATCH(h, { K::pack_all_table(h); return 0; });
        return h->fail(RNNWF_ERR_INVALID, "no stacked-layer gradient for NFULL=%d", h->NFULL);
    }
    GRAD_DISPATCH(h, { K::template pack_bwd<Lin>(h); return 0; });
    return h->fail(RNNWF_ERR_INVALID, "no gradient kernel for NFULL=%d", h->NFULL);
}
// Where every entry of the flat gradient (order and shapes of rnnwf_get_grads_flat) sits in the dW image: the host unpacker run on an
// image whose element k holds k + 1 - sidx[j] = +-(k + 1), 0: no source (stays 0).  is_f64: element type of the image.
int rnnwf::grad_flat_probe(rnnwf_handle* h, std::vector<int32_t>& sidx, size_t* dw_count, bool* is_f64) {
    const bool cplx = h->model == RNNWF_MODEL_CRNN_U1;
    const bool f64 = h->model == RNNWF_MODEL_GRU1D_F64;
    int pcols = 0, qcols = 0, hgn = 0;
    size_t n = 0;
    const auto saved = h->grads;
    if (h->NL != 1) {
        bool done = false;
        MLGRAD_DISPATCH(h, { if (K::DW_FLOATS < ((size_t)1 << 24)) { K::probe_unpack(h, &n); done = true; } break; });
        if (!done) return h->fail(RNNWF_ERR_INVALID, "stacked gradient image cannot be probed");
    } else {
    GRAD_DISPATCH(h, { pcols = K::G::PCOLS; qcols = K::G::QCOLS; hgn = K::G::HEAD_ROW * (cplx ? 3 : 1); break; });
    n = (size_t)pcols * qcols + hgn;
    if (n == 0 || n >= ((size_t)1 << 24)) return h->fail(RNNWF_ERR_INVALID, "gradient image of %zu elements cannot be probed", n);
    }
    if (h->NL != 1) {
    } else if (f64) {
        std::vector<double> img(n);
        for (size_t k = 0; k < n; ++k) img[k] = (double)(k + 1);
        GRAD_DISPATCH(h, { K::unpack(h, img.data(), (size_t)pcols * qcols); break; });
    } else {
        std::vector<float> img(n);
        for (size_t k = 0; k < n; ++k) img[k] = (float)(k + 1);
        GRAD_DISPATCH(h, { K::unpack(h, img.data(), (size_t)pcols * qcols); break; });
    }
    sidx.clear();
    for (auto& kv : h->params) {
        auto it = h->grads.find(kv.first);
        if (it == h->grads.end() || it->second.size() != kv.second.value.size()) {
            h->grads = saved;
            return h->fail(RNNWF_ERR_STATE, "grad_flat_probe: no gradient for '%s'", kv.first.c_str());
        }
        for (size_t i = 0; i < kv.second.slot.size(); ++i) sidx.push_back((int32_t)std::llround(it->second[(size_t)kv.second.slot[i]]));
    }
    h->grads = saved;
    if (dw_count) *dw_count = n;
    if (is_f64) *is_f64 = f64;
    return 0;
}

extern "C" int rnnwf_get_grad(rnnwf_handle* h, const char* name, void* data, int64_t count, int32_t dtype) {
    if (!h || !name || !data) return RNNWF_ERR_INVALID;
    auto it = h->grads.find(name);
    if (it == h->grads.end()) return h->fail(RNNWF_ERR_STATE, "no gradient for '%s' (call rnnwf_vmc_gradient first)", name);
    auto ps = h->params.find(name);                  // the gradient is computed for the padded parameter; the caller gets its own shape
    if (ps == h->params.end() || it->second.size() != ps->second.value.size())
        return h->fail(RNNWF_ERR_STATE, "gradient '%s' does not match a parameter of this model", name);
    const std::vector<int64_t>& slot = ps->second.slot;
    if ((int64_t)slot.size() != count)
        return h->fail(RNNWF_ERR_INVALID, "gradient '%s' has %lld elements, caller passed %lld", name,
                       (long long)slot.size(), (long long)count);
    if (dtype == RNNWF_F32) for (int64_t i = 0; i < count; ++i) ((float*)data)[i] = (float)it->second[slot[i]];
    else if (dtype == RNNWF_F64) for (int64_t i = 0; i < count; ++i) ((double*)data)[i] = it->second[slot[i]];
    else return h->fail(RNNWF_ERR_INVALID, "unknown dtype %d", dtype);
    return RNNWF_OK;
}

// All gradients in ONE call, in the order and shapes of rnnwf_set_params_flat.
extern "C" int rnnwf_get_grads_flat(rnnwf_handle* h, double* flat, int64_t count) {
    if (!h || !flat) return RNNWF_ERR_INVALID;
    if (h->grads.empty()) return h->fail(RNNWF_ERR_STATE, "rnnwf_get_grads_flat: no gradients (call rnnwf_vmc_gradient first)");
    int64_t total = 0;
    for (auto& kv : h->params) total += (int64_t)kv.second.slot.size();
    if (count != total)
        return h->fail(RNNWF_ERR_INVALID, "rnnwf_get_grads_flat: the model has %lld parameters, caller passed %lld", (long long)total, (long long)count);
    int64_t off = 0;
    for (auto& kv : h->params) {
        auto it = h->grads.find(kv.first);
        if (it == h->grads.end() || it->second.size() != kv.second.value.size())
            return h->fail(RNNWF_ERR_STATE, "rnnwf_get_grads_flat: no gradient for '%s'", kv.first.c_str());
        const std::vector<int64_t>& slot = kv.second.slot;
        for (size_t i = 0; i < slot.size(); ++i) flat[off + (int64_t)i] = it->second[slot[i]];
        off += (int64_t)slot.size();
    }
    return RNNWF_OK;
}

// the weight image changed: the backward image must be rebuilt on the next gradient call
void rnnwf::grad_invalidate(rnnwf_handle* h) { h->wbwd_valid = false; }     // (the buffer stays: freeing and re-allocating it cost ~0.1 ms per training iteration)
