// pack_split.h - host-side packing of the GRU parameters into the bf16x3 image of split_core.h.
#pragma once
#include <cmath>
#include <cstring>

#include "pack.h"
#include "split_core.h"
#include "split16_core.h"

namespace rnnwf {

// Image of the 16x16x32 form of the flip pass at 37..52 units (split16_core.h: S16nLayout).
template <int NOUT>
std::vector<char> pack_split16n_image(const rnnwf_handle* h) {
    using L = S16nLayout<NOUT>;
    static_assert(NOUT == 1, "the 16x16x32 form carries one head row (positive RNN)");
    const int H = h->H;
    std::vector<char> img(L::BYTES, 0);
    const std::string pre = kGruPre;
    const auto& Wg = pv(h, pre + "gates/kernel");
    const auto& bg = pv(h, pre + "gates/bias");
    const auto& Wci = pv(h, pre + "candidate/input_projection/kernel");
    const auto& bci = pv(h, pre + "candidate/input_projection/bias");
    const auto& Wch = pv(h, pre + "candidate/hidden_projection/kernel");
    const auto& bch = pv(h, pre + "candidate/hidden_projection/bias");
    const auto& Wd = pv(h, "wf_dense/kernel");
    const auto& bd = pv(h, "wf_dense/bias");
    const double sg = PackScale<float>::gate, sc = PackScale<float>::cand;
    // row (tile t, row i = 4 g + r) -> (gate, unit): gate 0 r, 1 u, 2 candidate, 3 head; unit -1: unused row
    auto decode = [&](int t, int g, int r, int& gate, int& unit) {
        if (t < 6) { gate = t / 2; unit = 4 * (4 * (t % 2) + r) + g; }
        else if (t < 9) { gate = t - 6; unit = 4 * (8 + r) + g; }
        else if (r < 3) { gate = r; unit = 4 * 12 + g; }
        else { gate = 3; unit = 0; }
        if (gate < 3 && unit >= H) unit = -1;
    };
    auto weight = [&](int gate, int uo, int ui) -> double {
        if (ui >= H) return 0.0;
        return gate == 0 ? sg * Wg[(size_t)(2 + ui) * 2 * H + uo]
             : gate == 1 ? sg * Wg[(size_t)(2 + ui) * 2 * H + H + uo]
             : gate == 2 ? sc * Wch[(size_t)ui * H + uo]
                         : Wd[(size_t)ui * 2 + 1] - Wd[(size_t)ui * 2];
    };
    uint16_t* A = reinterpret_cast<uint16_t*>(img.data() + L::OFF_A);
    float* CI = reinterpret_cast<float*>(img.data() + L::OFF_CI);
    for (int t = 0; t < L::NT; ++t)
        for (int i = 0; i < 16; ++i) {
            const int go = i >> 2, r = i & 3;
            int gate, uo;
            decode(t, go, r, gate, uo);
            if (uo < 0) continue;
            for (int sgm = 0; sgm < 2; ++sgm) {                  // accumulator start value: bias + one-hot input row
                const double v = gate == 0 ? sg * (bg[uo] + Wg[(size_t)sgm * 2 * H + uo])
                               : gate == 1 ? sg * (bg[H + uo] + Wg[(size_t)sgm * 2 * H + H + uo])
                               : gate == 2 ? sc * bch[uo] : bd[1] - bd[0];
                CI[(((size_t)sgm * L::NT + t) * 4 + go) * 4 + r] = (float)v;
            }
            for (int g = 0; g < 4; ++g) {                        // K side: lane group g supplies its units 4 j + g
                const int lane = (g << 4) | i;
                auto frag = [&](int f) { return A + (((size_t)t * L::NFR + f) * 64 + lane) * 8; };
                for (int e = 0; e < 8; ++e) {                    // octet j = 0..7: fragment a = weight part a
                    uint16_t p[3];
                    split3(weight(gate, uo, 4 * e + g), p);
                    for (int a = 0; a < 3; ++a) frag(a)[e] = p[a];
                }
                for (int e = 0; e < 4; ++e) {                    // j = 8..11: two products per k-step (entries e | 4 + e)
                    uint16_t p[3];
                    split3(weight(gate, uo, 4 * (8 + e) + g), p);
                    frag(3)[e] = p[0]; frag(3)[4 + e] = p[0];    // (w1 | w1) x (h2 | h1)
                    frag(4)[e] = p[1]; frag(4)[4 + e] = p[0];    // (w2 | w1) x (h1 | h3)
                    frag(5)[e] = p[1]; frag(5)[4 + e] = p[2];    // (w2 | w3) x (h2 | h1)
                }
                uint16_t p[3];
                split3(weight(gate, uo, 4 * 12 + g), p);
                const int part[6] = {0, 0, 0, 1, 1, 2};          // against the B entries {h1, h2, h3, h1, h2, h1}
                for (int e = 0; e < 6; ++e) frag(6)[e] = p[part[e]];
            }
        }
    float* XC = reinterpret_cast<float*>(img.data() + L::OFF_XC);
    float* WD = reinterpret_cast<float*>(img.data() + L::OFF_WD);
    float* BD = reinterpret_cast<float*>(img.data() + L::OFF_BD);
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < L::NJ; ++j) {
            const int u = 4 * j + g;
            if (u >= H) continue;
            for (int sgm = 0; sgm < 2; ++sgm) XC[((size_t)sgm * 4 + g) * L::XCP + j] = (float)(sc * (bci[u] + Wci[(size_t)sgm * H + u]));
            WD[(size_t)g * L::XCP + j] = (float)(Wd[(size_t)u * 2 + 1] - Wd[(size_t)u * 2]);
        }
    BD[0] = (float)(bd[1] - bd[0]);
    return img;
}

// Image of the 16x16x32 form of the flip pass at 69..100 units (split16_core.h: S16Layout).
template <int NOUT, class S = double>
std::vector<char> pack_split16_image(const rnnwf_handle* h) {
    using L = S16Layout<NOUT>;
    using Out = PackSink<S>;
    static_assert(NOUT == 1, "the 16x16x32 form carries one head row (positive RNN)");
    const int H = h->H;
    std::vector<char> img(L::BYTES, 0);
    Out::begin(img);
    const std::string pre = kGruPre;
    const auto Wg = pvs<S>(h, pre + "gates/kernel");
    const auto bg = pvs<S>(h, pre + "gates/bias");
    const auto Wci = pvs<S>(h, pre + "candidate/input_projection/kernel");
    const auto bci = pvs<S>(h, pre + "candidate/input_projection/bias");
    const auto Wch = pvs<S>(h, pre + "candidate/hidden_projection/kernel");
    const auto bch = pvs<S>(h, pre + "candidate/hidden_projection/bias");
    const auto Wd = pvs<S>(h, "wf_dense/kernel");
    const auto bd = pvs<S>(h, "wf_dense/bias");
    const double sg = PackScale<float>::gate, sc = PackScale<float>::cand;
    // row (tile t, row i = 4 g + r) -> (gate, unit): gate 0 r, 1 u, 2 candidate, 3 head; unit -1: unused row
    auto decode = [&](int t, int g, int r, int& gate, int& unit) {
        if (t < L::NTF) {
            const int b = t / 6, tau = t % 2;
            gate = (t % 6) / 2;
            unit = 4 * (8 * b + 4 * tau + r) + g;
        } else if (r < 3) {
            gate = r;
            unit = 4 * L::NJA + g;
        } else {
            gate = 3;
            unit = 0;
        }
        if (gate < 3 && unit >= H) unit = -1;
    };
    auto weight = [&](int gate, int uo, int ui) -> S {
        if (ui >= H) return S(0.0);
        if (gate == 0) return sg * Wg[(size_t)(2 + ui) * 2 * H + uo];
        if (gate == 1) return sg * Wg[(size_t)(2 + ui) * 2 * H + H + uo];
        if (gate == 2) return sc * Wch[(size_t)ui * H + uo];
        return Wd[(size_t)ui * 2 + 1] - Wd[(size_t)ui * 2];
    };
    uint16_t* A = reinterpret_cast<uint16_t*>(img.data() + L::OFF_A);
    uint16_t* ASP = reinterpret_cast<uint16_t*>(img.data() + L::OFF_ASP);
    float* CI = reinterpret_cast<float*>(img.data() + L::OFF_CI);
    for (int t = 0; t < L::NT; ++t)
        for (int i = 0; i < 16; ++i) {
            const int go = i >> 2, r = i & 3;
            int gate, uo;
            decode(t, go, r, gate, uo);
            if (uo < 0) continue;
            for (int sgm = 0; sgm < 2; ++sgm) {                  // accumulator start value: bias + one-hot input row
                S v;
                if (gate == 0) v = sg * (bg[uo] + Wg[(size_t)sgm * 2 * H + uo]);
                else if (gate == 1) v = sg * (bg[H + uo] + Wg[(size_t)sgm * 2 * H + H + uo]);
                else if (gate == 2) v = sc * bch[uo];
                else v = bd[1] - bd[0];
                Out::put(&CI[(((size_t)sgm * L::NT + t) * 4 + go) * 4 + r], v);
            }
            for (int g = 0; g < 4; ++g) {                        // K side: lane group g supplies its units 4 j + g
                const int lane = (g << 4) | i;
                for (int o = 0; o < L::NOCT; ++o)
                    for (int e = 0; e < 8; ++e) {
                        auto at = [&](int a) { return &A[((((size_t)t * 3 + a) * L::NOCT + o) * 64 + lane) * 8 + e]; };
                        Out::put_parts(at(0), at(1), at(2), weight(gate, uo, 4 * (8 * o + e) + g));
                    }
                const S ws = weight(gate, uo, 4 * L::NJA + g);
                const int part[6] = {0, 0, 0, 1, 1, 2};          // against the B entries {h1, h2, h3, h1, h2, h1}
                for (int e = 0; e < 6; ++e) Out::put_part(&ASP[((size_t)t * 64 + lane) * 8 + e], ws, part[e]);
            }
        }
    float* XC = reinterpret_cast<float*>(img.data() + L::OFF_XC);
    float* WD = reinterpret_cast<float*>(img.data() + L::OFF_WD);
    float* BD = reinterpret_cast<float*>(img.data() + L::OFF_BD);
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < L::NJ; ++j) {
            const int u = 4 * j + g;
            if (u >= H) continue;
            for (int sgm = 0; sgm < 2; ++sgm) Out::put(&XC[((size_t)sgm * 4 + g) * L::XCP + j], sc * (bci[u] + Wci[(size_t)sgm * H + u]));
            Out::put(&WD[(size_t)g * L::XCP + j], Wd[(size_t)u * 2 + 1] - Wd[(size_t)u * 2]);
        }
    Out::put(&BD[0], bd[1] - bd[0]);
    return img;
}

// bf16x3 A fragments of the cooperative base pass (layout.h: BaseBfLayout): the rows of pack_gru_image (same scaled f32
// weights), each split exactly into three bf16 parts.
template <int NFULL, class S = double>
std::vector<char> pack_base_bf_image(const rnnwf_handle* h) {
    using B = BaseBfLayout<NFULL>;
    using Out = PackSink<S>;
    const int H = h->H;
    std::vector<char> img(B::BYTES, 0);
    Out::begin(img);
    const std::string pre = kGruPre;
    const auto Wg = pvs<S>(h, pre + "gates/kernel");
    const auto Wch = pvs<S>(h, pre + "candidate/hidden_projection/kernel");
    const double sg = PackScale<float>::gate, sc = PackScale<float>::cand;
    auto wt = [&](int gate, int unit, int k) -> S {
        if (k >= H) return S(0.0);
        if (gate == 0) return sg * Wg[(size_t)(2 + k) * 2 * H + unit];
        if (gate == 1) return sg * Wg[(size_t)(2 + k) * 2 * H + H + unit];
        return sc * Wch[(size_t)k * H + unit];
    };
    uint16_t* A = reinterpret_cast<uint16_t*>(img.data());
    for (int tile = 0; tile < B::NT; ++tile)
        for (int row = 0; row < 16; ++row) {
            const int q = row >> 2, r = row & 3;                // C/D row 4 q + r of the f32 16x16 output
            int gate, unit;
            if (tile < 3 * NFULL) { gate = tile / NFULL; unit = 16 * (tile % NFULL) + 4 * r + q; }
            else { if (r == 3) continue; gate = r; unit = 16 * NFULL + q; }
            if (unit >= H) continue;
            for (int t = 0; t < B::NKS; ++t)
                for (int g = 0; g < 4; ++g)
                    for (int e = 0; e < 8; ++e) {
                        const int kappa = 32 * t + 8 * g + e, grp = kappa / 16, loc = kappa % 16, kq = loc / 4, kr = loc % 4;
                        int ku = -1;
                        if (grp < NFULL) ku = 16 * grp + 4 * kr + kq;
                        else if (grp == NFULL && kr == 0) ku = 16 * NFULL + kq;
                        if (ku < 0 || ku >= H) continue;
                        const int lane = (g << 4) | row;
                        auto at = [&](int a) { return &A[((((size_t)tile * 3 + a) * B::NKS + t) * 64 + lane) * 8 + e]; };
                        Out::put_parts(at(0), at(1), at(2), wt(gate, unit, ku));      // (the split starts from the value as f32)
                    }
        }
    return img;
}

template <int NF32, int RJ, int NOUT = 1, int MODE = 0, class S = double>
std::vector<char> pack_split_image(const rnnwf_handle* h) {
    using L = SplitLayout<NF32, RJ, NOUT, MODE>;
    using Out = PackSink<S>;
    const int H = h->H;
    std::vector<char> img(L::BYTES, 0);
    Out::begin(img);
    const std::string pre = kGruPre;
    const auto Wg = pvs<S>(h, pre + "gates/kernel");
    const auto bg = pvs<S>(h, pre + "gates/bias");
    const auto Wci = pvs<S>(h, pre + "candidate/input_projection/kernel");
    const auto bci = pvs<S>(h, pre + "candidate/input_projection/bias");
    const auto Wch = pvs<S>(h, pre + "candidate/hidden_projection/kernel");
    const auto bch = pvs<S>(h, pre + "candidate/hidden_projection/bias");
    const auto Wd = pvs<S>(h, NOUT == 1 ? "wf_dense/kernel" : "wf_dense_ampl/kernel");
    const auto bd = pvs<S>(h, NOUT == 1 ? "wf_dense/bias" : "wf_dense_ampl/bias");
    const double sg = PackScale<float>::gate, sc = PackScale<float>::cand;
    uint16_t* A = reinterpret_cast<uint16_t*>(img.data() + L::OFF_A);
    auto head_w = [&](int o, int ui) -> S {
        if (o == 0) return Wd[(size_t)ui * 2 + 1] - Wd[(size_t)ui * 2];
        return pvs<S>(h, "wf_dense_phase/kernel")[(size_t)ui * 2 + (o - 1)];
    };
    auto head_b = [&](int o) -> S {
        if (o == 0) return bd[1] - bd[0];
        return pvs<S>(h, "wf_dense_phase/bias")[o - 1];
    };
    for (int T = 0; T < L::NT; ++T)
        for (int lane = 0; lane < 64; ++lane) {
            const int r32 = lane & 31, hhk = lane >> 5;
            const int hh_row = (r32 >> 2) & 1, rho = (r32 & 3) + 4 * (r32 >> 3);
            int g = -1, uo = -1;
            if (T < 3 * NF32) {
                g = T % 3;
                uo = L::unit_of(16 * (T / 3) + rho, hh_row);
            } else {
                const int s = 16 * (T - 3 * NF32) + rho;
                if (s < 3 * RJ) { g = s / RJ; uo = L::unit_of(16 * NF32 + s % RJ, hh_row); }
                // modes 2 and 3: the spare slots behind the remainder units carry the HEAD rows
                // (g = 3 + o, both lane halves the same row): their kernels read the head of the state that ENTERED a step
                // from its accumulators
                else if ((MODE == 2 || MODE == 3) && s < L::HEAD_SLOT + NOUT) { g = 3 + (s - L::HEAD_SLOT); uo = 0; }
            }
            if (g < 0 || uo >= H) continue;
            auto weight = [&](int ui) -> S {               // (scaled) recurrent weight from unit ui into row (g, uo)
                if (ui >= H) return S(0.0);
                if (g == 0) return sg * Wg[(size_t)(2 + ui) * 2 * H + uo];
                if (g == 1) return sg * Wg[(size_t)(2 + ui) * 2 * H + H + uo];
                if (g == 2) return sc * Wch[(size_t)ui * H + uo];
                return head_w(g - 3, ui);
            };
            if constexpr (MODE != 0) {
                // accumulator start value of this row: bias + one-hot input row (the same for both K halves)
                if (hhk == 0) {
                    float* CI = reinterpret_cast<float*>(img.data() + L::OFF_CI);
                    for (int sgm = 0; sgm < 2; ++sgm) {
                        S v;
                        if (g == 0) v = sg * (bg[uo] + Wg[(size_t)sgm * 2 * H + uo]);
                        else if (g == 1) v = sg * (bg[H + uo] + Wg[(size_t)sgm * 2 * H + H + uo]);
                        else if (g == 2) v = sc * bch[uo];
                        else v = head_b(g - 3);
                        Out::put(&CI[(((size_t)sgm * L::NT + T) * 2 + hh_row) * 16 + rho], v);
                    }
                }
            }
            if constexpr (MODE == 1) {
                // one chain over the concatenated K axis: lane half hhk's flat register list, product by product
                constexpr int ORD[6][2] = {{2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};
                for (int ks = 0; ks < L::KS; ++ks)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int fe = 8 * ks + jj, f = fe / 2;
                        if (f >= 6 * L::NRM) continue;
                        const int o = f / L::NRM, e = 2 * (f % L::NRM) + (fe & 1);
                        Out::put_part(&A[(((size_t)ks * L::NT + T) * 64 + lane) * 8 + jj], weight(L::unit_of(e, hhk)), ORD[o][0]);
                    }
                continue;
            }
            if constexpr (MODE == 2) {
                // special unit of K half hhk: parts {w1, w1, w1, w2, w2, w3} against the B entries {h1, h2, h3, h1, h2, h1}
                const S ws = weight(L::unit_of(L::NU - 1, hhk));
                uint16_t* ASP = reinterpret_cast<uint16_t*>(img.data() + L::OFF_ASP);
                const int part[6] = {0, 0, 0, 1, 1, 2};
                for (int jj = 0; jj < 6; ++jj) Out::put_part(&ASP[((size_t)T * 64 + lane) * 8 + jj], ws, part[jj]);
            }
            if constexpr (MODE == 3) {
                // NS special units per K half: entry 6 s + i of the special k-steps = product i of special unit s,
                // A parts {w1, w1, w1, w2, w2, w3} against the B entries {h1, h2, h3, h1, h2, h1}
                uint16_t* ASP = reinterpret_cast<uint16_t*>(img.data() + L::OFF_ASP);
                const int part[6] = {0, 0, 0, 1, 1, 2};
                for (int sp = 0; sp < L::NS; ++sp) {
                    const S ws = weight(L::unit_of(L::NUA + sp, hhk));
                    for (int i = 0; i < 6; ++i) {
                        const int idx = 6 * sp + i;
                        Out::put_part(&ASP[(((size_t)T * L::KSP + idx / 8) * 64 + lane) * 8 + idx % 8], ws, part[i]);
                    }
                }
            }
            for (int x = 0; x < L::NQ; ++x)
                for (int jj = 0; jj < 8; ++jj) {
                    const int e = 8 * x + jj;
                    S w = S(0.0);
                    if (e < L::NUA) {
                        w = weight(L::unit_of(e, hhk));
                    } else if (MODE != 0) {
                        w = S(0.0);
                    } else if (e == L::NU && hhk == 0) {                    // bias (+ input row of spin 0)
                        if (g == 0) w = sg * (bg[uo] + Wg[uo]);
                        else if (g == 1) w = sg * (bg[H + uo] + Wg[H + uo]);
                        else if (g == 2) w = sc * bch[uo];
                        else w = head_b(g - 3);
                    } else if (e == L::NU + 1 && hhk == 0) {                // input row difference, times sigma
                        if (g == 0) w = sg * (Wg[(size_t)2 * H + uo] - Wg[uo]);
                        else if (g == 1) w = sg * (Wg[(size_t)2 * H + H + uo] - Wg[H + uo]);
                    }
                    auto at = [&](int a) { return &A[((((size_t)T * 3 + a) * L::NQ + x) * 64 + lane) * 8 + jj]; };
                    Out::put_parts(at(0), at(1), at(2), w);
                }
        }
    float* XC = reinterpret_cast<float*>(img.data() + L::OFF_XC);
    float* WD = reinterpret_cast<float*>(img.data() + L::OFF_WD);
    float* BD = reinterpret_cast<float*>(img.data() + L::OFF_BD);
    for (int hh = 0; hh < 2; ++hh)
        for (int e = 0; e < L::NU; ++e) {
            const int u = L::unit_of(e, hh);
            if (u >= H) continue;
            for (int sgm = 0; sgm < 2; ++sgm)
                Out::put(&XC[(size_t)(sgm * 2 + hh) * L::NUP + e], sc * (bci[u] + Wci[(size_t)sgm * H + u]));
            for (int o = 0; o < NOUT; ++o) Out::put(&WD[((size_t)hh * L::NUP + e) * NOUT + o], head_w(o, u));
        }
    for (int o = 0; o < NOUT; ++o) Out::put(&BD[o], head_b(o));
    return img;
}

// Image of GRU layer `layer` >= 1 on the bf16x3 engine (split_core.h: SplitUpperLayout): the X block (input = state of the layer
// below) and the H block, each laid out as the MODE 2 image of the first layer, the biases as accumulator start values, and - for
// the top layer - the head rows in the H block's spare mixed-tile slots plus the VALU head's tables.
template <int NF32, int RJ, int NOUT, class S = double>
std::vector<char> pack_split_upper_image(const rnnwf_handle* h, int layer, bool top) {
    using U = SplitUpperLayout<NF32, RJ, NOUT>;
    using L = typename U::L0;
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
    const auto Wd = pvs<S>(h, NOUT == 1 ? "wf_dense/kernel" : "wf_dense_ampl/kernel");
    const auto bd = pvs<S>(h, NOUT == 1 ? "wf_dense/bias" : "wf_dense_ampl/bias");
    const double sg = PackScale<float>::gate, sc = PackScale<float>::cand;
    auto head_w = [&](int o, int ui) -> S {
        if (o == 0) return Wd[(size_t)ui * 2 + 1] - Wd[(size_t)ui * 2];
        return pvs<S>(h, "wf_dense_phase/kernel")[(size_t)ui * 2 + (o - 1)];
    };
    auto head_b = [&](int o) -> S {
        if (o == 0) return bd[1] - bd[0];
        return pvs<S>(h, "wf_dense_phase/bias")[o - 1];
    };
    for (int blk = 0; blk < 2; ++blk) {
        const bool xb = blk == 0;
        uint16_t* A = reinterpret_cast<uint16_t*>(img.data() + (xb ? U::OFF_AX : U::OFF_AH));
        uint16_t* ASP = reinterpret_cast<uint16_t*>(img.data() + (xb ? U::OFF_AX : U::OFF_AH) + L::OFF_ASP);
        for (int T = 0; T < L::NT; ++T)
            for (int lane = 0; lane < 64; ++lane) {
                const int r32 = lane & 31, hhk = lane >> 5;
                const int hh_row = (r32 >> 2) & 1, rho = (r32 & 3) + 4 * (r32 >> 3);
                int g = -1, uo = -1;                            // g: 0 r, 1 u, 2 candidate (X: y, H: q), 3 + o: head row o (H block, top layer)
                if (T < 3 * NF32) {
                    g = T % 3;
                    uo = L::unit_of(16 * (T / 3) + rho, hh_row);
                } else {
                    const int sl = 16 * (T - 3 * NF32) + rho;
                    if (sl < 3 * RJ) { g = sl / RJ; uo = L::unit_of(16 * NF32 + sl % RJ, hh_row); }
                    else if (!xb && top && sl < L::HEAD_SLOT + NOUT) { g = 3 + (sl - L::HEAD_SLOT); uo = 0; }
                }
                if (g < 0 || uo >= H) continue;
                auto weight = [&](int ui) -> S {               // (scaled) weight from input unit ui into row (g, uo)
                    if (ui >= H) return S(0.0);
                    const size_t row = xb ? (size_t)ui : (size_t)(H + ui);
                    if (g == 0) return sg * Wg[row * 2 * H + uo];
                    if (g == 1) return sg * Wg[row * 2 * H + H + uo];
                    if (g == 2) return xb ? sc * Wci[(size_t)ui * H + uo] : sc * Wch[(size_t)ui * H + uo];
                    return head_w(g - 3, ui);
                };
                {   // special unit of K half hhk: parts {w1, w1, w1, w2, w2, w3} against the B entries {h1, h2, h3, h1, h2, h1}
                    const S ws = weight(L::unit_of(L::NU - 1, hhk));
                    const int part[6] = {0, 0, 0, 1, 1, 2};
                    for (int jj = 0; jj < 6; ++jj) Out::put_part(&ASP[((size_t)T * 64 + lane) * 8 + jj], ws, part[jj]);
                    // K entries 6, 7 of both halves meet the constant 1.0: the three parts of the row's bias.  The r / u biases enter
                    // once, through the X block; the H block carries the candidate's hidden bias and the head biases.
                    S bias = S(0.0);
                    if (g == 0) { if (xb) bias = sg * bg[uo]; }
                    else if (g == 1) { if (xb) bias = sg * bg[H + uo]; }
                    else if (g == 2) bias = xb ? sc * bci[uo] : sc * bch[uo];
                    else bias = head_b(g - 3);
                    uint16_t* b6 = &ASP[((size_t)T * 64 + lane) * 8 + 6];
                    uint16_t* b7 = &ASP[((size_t)T * 64 + lane) * 8 + 7];
                    if (hhk == 0) { Out::put_part(b6, bias, 0); Out::put_part(b7, bias, 1); }
                    else { Out::put_part(b6, bias, 2); }
                }
                for (int x = 0; x < L::NQ; ++x)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int e = 8 * x + jj;
                        auto at = [&](int a) { return &A[((((size_t)T * 3 + a) * L::NQ + x) * 64 + lane) * 8 + jj]; };
                        Out::put_parts(at(0), at(1), at(2), e < L::NUA ? weight(L::unit_of(e, hhk)) : S(0.0));
                    }
            }
    }
    if (top) {
        float* WD = reinterpret_cast<float*>(img.data() + U::OFF_WD);
        float* BD = reinterpret_cast<float*>(img.data() + U::OFF_BD);
        for (int hh = 0; hh < 2; ++hh)
            for (int e = 0; e < L::NU; ++e) {
                const int u = L::unit_of(e, hh);
                if (u >= H) continue;
                for (int o = 0; o < NOUT; ++o) Out::put(&WD[((size_t)hh * L::NUP + e) * NOUT + o], head_w(o, u));
            }
        for (int o = 0; o < NOUT; ++o) Out::put(&BD[o], head_b(o));
    }
    return img;
}

}  // namespace rnnwf
