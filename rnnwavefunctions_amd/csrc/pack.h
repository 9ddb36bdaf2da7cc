// pack.h - host-side packing of TF-named parameters into the LDS weight image of layout.h.
#pragma once
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "handle.h"
#include "layout.h"
#include "pack_value.h"

namespace rnnwf {

static const char* kGruPre = "multi_rnn_cell/cell_0/cudnn_compatible_gru_cell/";

inline const std::vector<double>& pv(const rnnwf_handle* h, const std::string& name) {
    return h->params.at(name).value;
}
// the same tensor for a packer written over the scalar type S (pack_value.h): values, or values with their provenance
template <class S> inline ParamView<S> pvs(const rnnwf_handle* h, const std::string& name);
template <> inline ParamView<double> pvs<double>(const rnnwf_handle* h, const std::string& name) {
    return ParamView<double>{&h->params.at(name).value};
}
template <> inline ParamView<Lin> pvs<Lin>(const rnnwf_handle* h, const std::string& name) {
    return ParamView<Lin>{&h->params.at(name).value, &h->param_flat.at(name)};
}

// Row scales that let the f32 kernels feed the accumulator straight into v_exp_f32 (see gru_core.h, Act<T>).
template <typename T> struct PackScale;
template <> struct PackScale<float> {
    static constexpr double gate = -1.44269504088896340736, cand = -2.88539008177792681472;
};
template <> struct PackScale<double> {
    static constexpr double gate = 1.0, cand = 1.0;
};

// Packs the single-layer cuDNN-compatible GRU + Dense head(s).
// NOUT = 1: positive RNN (head row = softmax logit difference z1 - z0);
// NOUT = 3: complex RNN (amplitude logit difference, phase logit 0, phase logit 1).
template <typename T, int NFULL, int NOUT, class S = double>
std::vector<char> pack_gru_image(const rnnwf_handle* h) {
    using L = GruLayout<T, NFULL, NOUT>;
    using Out = PackSink<S>;
    const int H = h->H;
    std::vector<char> img(L::BYTES, 0);
    Out::begin(img);
    const std::string pre = kGruPre;
    const auto Wg = pvs<S>(h, pre + "gates/kernel");                         // [2+H, 2H], cols r | u
    const auto bg = pvs<S>(h, pre + "gates/bias");                           // [2H]
    const auto Wci = pvs<S>(h, pre + "candidate/input_projection/kernel");   // [2, H]
    const auto bci = pvs<S>(h, pre + "candidate/input_projection/bias");     // [H]
    const auto Wch = pvs<S>(h, pre + "candidate/hidden_projection/kernel");  // [H, H]
    const auto bch = pvs<S>(h, pre + "candidate/hidden_projection/bias");    // [H]

    auto decode = [&](int tile, int q, int r, int& gate, int& unit) -> bool {
        if (tile < 3 * NFULL) {
            gate = tile / NFULL;
            unit = 16 * (tile % NFULL) + 4 * r + q;
        } else {
            if (r == 3) return false;
            gate = r;
            unit = 16 * NFULL + q;
        }
        return unit < H;
    };
    const double sg = PackScale<T>::gate, sc = PackScale<T>::cand;
    auto wt = [&](int gate, int unit, int k) -> S {  // (scaled) W^T[row(gate,unit)][k]
        if (k >= H) return S(0.0);
        if (gate == 0) return sg * Wg[(size_t)(2 + k) * 2 * H + unit];
        if (gate == 1) return sg * Wg[(size_t)(2 + k) * 2 * H + H + unit];
        return sc * Wch[(size_t)k * H + unit];
    };
    T* avec = reinterpret_cast<T*>(img.data() + L::OFF_AVEC);
    T* arem = reinterpret_cast<T*>(img.data() + L::OFF_AREM);
    for (int tile = 0; tile < L::NT; ++tile)
        for (int row = 0; row < 16; ++row) {
            int q, r, gate, unit;
            row_to_qr<T>(row, q, r);
            if (!decode(tile, q, r, gate, unit)) continue;
            for (int kq = 0; kq < 4; ++kq) {       // lane quarter of the A operand = k mod 4
                const int lane = (kq << 4) | row;
                for (int g = 0; g < L::NG; ++g)
                    for (int j = 0; j < L::VW; ++j)
                        Out::put(&avec[(((size_t)tile * L::NG + g) * 64 + lane) * L::VW + j], wt(gate, unit, 4 * (g * L::VW + j) + kq));
                Out::put(&arem[(size_t)tile * 64 + lane], wt(gate, unit, 4 * (L::KT - 1) + kq));
            }
        }
    for (int v = 0; v < 3; ++v) {  // v = 0: zero input; v = 1, 2: one-hot of spin 0, 1
        T* binit = reinterpret_cast<T*>(img.data() + L::OFF_BINIT + v * L::SZ_BINIT_VARIANT);
        T* xc = reinterpret_cast<T*>(img.data() + L::OFF_XC + v * L::SZ_XC_VARIANT);
        for (int tile = 0; tile < L::NT; ++tile)
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < 4; ++r) {
                    int gate, unit;
                    if (!decode(tile, q, r, gate, unit)) continue;
                    S b;
                    if (gate == 0) b = sg * (bg[unit] + (v ? Wg[(size_t)(v - 1) * 2 * H + unit] : S(0.0)));
                    else if (gate == 1) b = sg * (bg[H + unit] + (v ? Wg[(size_t)(v - 1) * 2 * H + H + unit] : S(0.0)));
                    else b = sc * bch[unit];
                    Out::put(&binit[tile * 16 + q * 4 + r], b);
                }
        for (int m = 0; m <= NFULL; ++m)
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < 4; ++r) {
                    if (m == NFULL && r != 0) continue;
                    const int unit = m < NFULL ? 16 * m + 4 * r + q : 16 * NFULL + q;
                    if (unit >= H) continue;
                    Out::put(&xc[m * 16 + q * 4 + r], sc * (bci[unit] + (v ? Wci[(size_t)(v - 1) * H + unit] : S(0.0))));
                }
    }
    T* wd = reinterpret_cast<T*>(img.data() + L::OFF_WD);
    T* bd = reinterpret_cast<T*>(img.data() + L::OFF_BD);
    const auto Wd = pvs<S>(h, std::string(NOUT == 1 ? "wf_dense" : "wf_dense_ampl") + "/kernel");   // [H, 2]
    const auto bdv = pvs<S>(h, std::string(NOUT == 1 ? "wf_dense" : "wf_dense_ampl") + "/bias");    // [2]
    for (int kt = 0; kt < L::KT; ++kt)
        for (int q = 0; q < 4; ++q) {
            const int unit = 4 * kt + q;
            if (unit >= H) continue;
            Out::put(&wd[q * L::WD_Q + kt * NOUT], Wd[(size_t)unit * 2 + 1] - Wd[(size_t)unit * 2]);
            if (NOUT == 3) {
                const auto Wp = pvs<S>(h, "wf_dense_phase/kernel");
                Out::put(&wd[q * L::WD_Q + kt * NOUT + 1], Wp[(size_t)unit * 2]);
                Out::put(&wd[q * L::WD_Q + kt * NOUT + 2], Wp[(size_t)unit * 2 + 1]);
            }
        }
    Out::put(&bd[0], bdv[1] - bdv[0]);
    if (NOUT == 3) {
        const auto bp = pvs<S>(h, "wf_dense_phase/bias");
        Out::put(&bd[1], bp[0]);
        Out::put(&bd[2], bp[1]);
    }
    return img;
}

// exp / log tables of device.h (F64Tables): 2^(j/64), 1/c_j, log(c_j) with c_j = 1 + (j + 1/2)/64
inline void fill_f64_tables(double* t) {
    for (int j = 0; j < 64; ++j) {
        const double c = 1.0 + (j + 0.5) / 64.0;
        t[F64Tables::EXP2 + j] = std::exp2(j / 64.0);
        t[F64Tables::RCPC + j] = 1.0 / c;
        t[F64Tables::LOGC + j] = std::log(c);
    }
}

// Image of GRU layer `layer` >= 1 (input = state of the layer below, dimension H), UpperLayout<NFULL, T>.
template <int NFULL, typename T = float, class S = double>
std::vector<char> pack_upper_image(const rnnwf_handle* h, int layer) {
    using U = UpperLayout<NFULL, T>;
    using Out = PackSink<S>;
    const int H = h->H;
    std::vector<char> img(U::BYTES, 0);
    Out::begin(img);
    const std::string pre = "multi_rnn_cell/cell_" + std::to_string(layer) + "/cudnn_compatible_gru_cell/";
    const auto Wg = pvs<S>(h, pre + "gates/kernel");                         // [H + H, 2H]: input rows first, cols r | u
    const auto bg = pvs<S>(h, pre + "gates/bias");
    const auto Wci = pvs<S>(h, pre + "candidate/input_projection/kernel");   // [H, H]
    const auto bci = pvs<S>(h, pre + "candidate/input_projection/bias");
    const auto Wch = pvs<S>(h, pre + "candidate/hidden_projection/kernel");  // [H, H]
    const auto bch = pvs<S>(h, pre + "candidate/hidden_projection/bias");
    const double sg = PackScale<T>::gate, sc = PackScale<T>::cand;
    // gate ids: 0 r, 1 u, 2 q (hidden candidate), 3 y (input candidate)
    auto decode = [&](bool xblock, int tile, int q, int r, int& gate, int& unit) -> bool {
        if (tile < 3 * NFULL) {
            gate = tile / NFULL;
            if (gate == 2 && xblock) gate = 3;
            unit = 16 * (tile % NFULL) + 4 * r + q;
        } else {
            if (r == (xblock ? 2 : 3)) return false;
            gate = r;
            unit = 16 * NFULL + q;
        }
        return unit < H;
    };
    auto wt = [&](bool xblock, int gate, int unit, int k) -> S {
        if (k >= H) return S(0.0);
        const size_t row = xblock ? k : H + k;
        if (gate == 0) return sg * Wg[row * 2 * H + unit];
        if (gate == 1) return sg * Wg[row * 2 * H + H + unit];
        if (gate == 2) return sc * Wch[(size_t)k * H + unit];
        return sc * Wci[(size_t)k * H + unit];
    };
    for (int blk = 0; blk < 2; ++blk) {
        const bool xb = blk == 0;
        T* avec = reinterpret_cast<T*>(img.data() + (xb ? U::OFF_AX : U::OFF_AH));
        T* arem = reinterpret_cast<T*>(img.data() + (xb ? U::OFF_AXR : U::OFF_AHR));
        for (int tile = 0; tile < U::NT; ++tile)
            for (int row = 0; row < 16; ++row) {
                int q, r, gate, unit;
                row_to_qr<T>(row, q, r);
                if (!decode(xb, tile, q, r, gate, unit)) continue;
                for (int kq = 0; kq < 4; ++kq) {
                    const int lane = (kq << 4) | row;
                    for (int g = 0; g < U::NG; ++g)
                        for (int j = 0; j < U::VW; ++j)
                            Out::put(&avec[(((size_t)tile * U::NG + g) * 64 + lane) * U::VW + j], wt(xb, gate, unit, 4 * (g * U::VW + j) + kq));
                    Out::put(&arem[(size_t)tile * 64 + lane], wt(xb, gate, unit, 4 * (U::KT - 1) + kq));
                }
            }
    }
    T* b = reinterpret_cast<T*>(img.data() + U::OFF_B);
    for (int t = 0; t < U::NT2; ++t)
        for (int q = 0; q < 4; ++q)
            for (int r = 0; r < 4; ++r) {
                const int gate = t < 4 * NFULL ? t / NFULL : r;
                const int unit = t < 4 * NFULL ? 16 * (t % NFULL) + 4 * r + q : 16 * NFULL + q;
                if (unit >= H) continue;
                S v;
                if (gate == 0) v = sg * bg[unit];
                else if (gate == 1) v = sg * bg[H + unit];
                else if (gate == 2) v = sc * bch[unit];
                else v = sc * bci[unit];
                Out::put(&b[t * 16 + q * 4 + r], v);
            }
    return img;
}

}  // namespace rnnwf
