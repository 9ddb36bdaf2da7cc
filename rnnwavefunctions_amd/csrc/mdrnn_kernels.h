// mdrnn_kernels.h - 2D vanilla RNN wave function on the zig-zag path, float64
// (2DTFIM_2DRNN/MDRNNcell.py:51-66, 2DTFIM_2DRNN/RNNwavefunction.py:35-200).
//
// One step:  h' = elu(x_h Uh + h_h Wh + x_v Uv + h_v Wv + b)  as  D^T = [Wh^T | Wv^T] [h_h ; h_v]  on the
// f64 16x16x4 MFMA (C/D row = q + 4 r, so the fragment order is the natural unit order and, as for the GRU,
// the output fragment is directly the next step's B operand).  One-hot inputs fold into the accumulator
// initialisation.  The units beyond the last full 16-row tile (2 of 50) are NOT padded to a fifth MFMA tile:
// on gfx950 the f64 MFMA and the f64 VALU have the same flop rate, so a tile with 2 live rows of 16 costs
// 26 MFMAs (1664 cycles) where 2 x 26 FMAs per lane plus three cross-quarter shuffles do the same work.  Hidden states of all sites live in HBM in fragment order:
//   hs [N][nsb][KP][64] double2  state after visit position p, k-steps in pairs (KP = ceil(KT/2); base pass: out;
//                                flip pass and gradient: in)
// because every site needs its vertical neighbour from the previous row.  A flip chain re-evaluates
// positions i+1..N-1; states it produces itself go to a private ring of 2*Nx positions per wave.
#pragma once
#include "device.h"

namespace rnnwf {

template <int NFULL_>
struct MdLayout {
    static constexpr int NFULL = NFULL_;
    static constexpr int KT = 4 * NFULL + 1;          // k-steps per hidden vector
    static constexpr int NT = NFULL + 1;              // 16-row output tiles (last: units 16 NFULL + q in reg 0)
    static constexpr size_t OFF_A = 0;                                       // [NFULL][KT][64] double2 (k-steps 2g, 2g+1 of [h_h ; h_v])
    static constexpr size_t OFF_WR = OFF_A + (size_t)NFULL * KT * 64 * 16;   // [2 KT][4 q][4 j] f64: row 4 kt + q of W -> remainder unit j
    static constexpr size_t OFF_BH = OFF_WR + (size_t)2 * KT * 16 * 8;       // [3][NT][4][4] f64 : b + Uh[x_h]
    static constexpr size_t SZ_B = ((size_t)NT * 16 + 4) * 8;
    static constexpr size_t OFF_BV = OFF_BH + 3 * SZ_B;                      // [3][NT][4][4] f64 : Uv[x_v]
    static constexpr size_t OFF_WD = OFF_BV + 3 * SZ_B;                      // [KT][4][2] f64
    static constexpr size_t OFF_BD = OFF_WD + (size_t)KT * 4 * 2 * 8;        // [2] f64
    static constexpr size_t OFF_WDD = OFF_BD + 16;                           // [KT][4] f64: Wd[:,1] - Wd[:,0], then bd[1] - bd[0]
    static constexpr size_t OFF_TAB = ((OFF_WDD + ((size_t)KT * 4 + 1) * 8 + 15) / 16) * 16;   // F64Tables (exp / log tables)
    static constexpr size_t OFF_BHV = OFF_TAB + F64Tables::BYTES;            // [3 x_h][3 x_v][NT][4][4] f64 : b + Uh[x_h] + Uv[x_v]
    static constexpr size_t BYTES = OFF_BHV + 9 * SZ_B;
    static constexpr size_t WORDS_BYTES = 8 * 64 * 4;                        // per wave, behind the image: the chain's spin words
};

struct MdArgs {
    const void* wimg;
    int32_t N, Nx;
    int32_t rem;                   // num_units - 16 NFULL (1..4): units computed on the VALU
    int64_t ns, nsb;
    uint32_t* bits;                // spins in visit order
    double* hs;                    // [N][nsb][KP][64] double2
    double* ring;                  // flip pass: [total waves][2 Nx][KP][64] double2 private states
    double* lpq;                   // [N+1][ns]
    double* out_lp;                // [ns]
    const int32_t* vert_pos;       // [N] visit position of the vertical neighbour, -1 at the first row
    const int32_t* row_first;      // [N] 1 if the site is the first visited of its row (no horizontal neighbour)
    const int32_t* row_of_pos;     // [N] lpq row of a flip at visit position p (= nx*Ny + ny + 1)
    uint64_t seed, step;
    int64_t sample_offset;
    int32_t sampling;
    int32_t ablate;                // diagnostics only (RNNWF_ABLATE), flip pass: 1 no ring stores, 2 no h_v loads, 4 no head,
                                   // 8 no elu, 16 no remainder-unit FMAs, 32 no MFMAs
    int64_t ntiles;
};

template <int NFULL>
struct MdCore {
    using L = MdLayout<NFULL>;
    using F = Frag<double>;
    using V4 = F::V4;
    typedef double V2 __attribute__((ext_vector_type(2)));
    static constexpr int KT = L::KT, NT = L::NT;

    static __device__ __forceinline__ void stage(char* lds, const void* wimg) {
        const uint4* src = reinterpret_cast<const uint4*>(wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (int i = threadIdx.x; i < (int)(L::BYTES / 16); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }

    static constexpr int KP = (KT + 1) / 2;           // a stored state: [KP][64] double2 (16-byte accesses; last pair padded)
    static __device__ __forceinline__ void load_state(const double* base, double* dst) {   // base includes + 2 * lane
        const V2* src = reinterpret_cast<const V2*>(base);
#pragma unroll
        for (int g = 0; g < KP; ++g) {
            const V2 v = src[g * 64];
            dst[2 * g] = v[0];
            if (2 * g + 1 < KT) dst[2 * g + 1] = v[1];
        }
    }
    static __device__ __forceinline__ void store_state(double* base, const double* srcv) {
        V2* dst = reinterpret_cast<V2*>(base);
#pragma unroll
        for (int g = 0; g < KP; ++g) dst[g * 64] = V2{srcv[2 * g], 2 * g + 1 < KT ? srcv[2 * g + 1] : 0.0};
    }

    // tf.nn.elu = max(x, 0) + (e^min(x, 0) - 1): no select, table-driven exp (abs error ~1e-16)
    static __device__ __forceinline__ double elu(double x, const double* tab) {
        return __builtin_fmax(x, 0.0) + (exp_tab(__builtin_fmin(x, 0.0), tab) - 1.0);
    }

    // hh = h_h fragment, hv = h_v fragment; result in out[KT] (may alias hh).  rem = num_units - 16 NFULL (1..4).
    // Everything that needs only h_h (its half of the remainder-unit FMAs, the first 2 floor(KT/2) k-steps of the MFMA
    // chain) is issued before anything touches h_v: in the flip pass h_v is a 7 KB read from HBM that has then ~2 300
    // cycles to land.
    struct NoFetch { __device__ __forceinline__ void operator()() const {} };
    static __device__ __forceinline__ void step(const char* lds, int sig_h, int sig_v, const double (&hh)[KT],
                                                const double (&hv)[KT], double (&out)[KT], int lane, int rem, int ablate = 0) {
        step_fetch(lds, sig_h, sig_v, hh, hv, out, lane, rem, ablate, NoFetch{});
    }
    // fetch_hv(): called once, after everything that needs only h_h has been issued and before the first use of hv - the
    // prefetching flip kernel fills hv there (from its LDS staging slot) and starts the transfer of the next step's h_v
    template <typename Fetch>
    static __device__ __forceinline__ void step_fetch(const char* lds, int sig_h, int sig_v, const double (&hh)[KT],
                                                      const double (&hv)[KT], double (&out)[KT], int lane, int rem, int ablate, Fetch&& fetch_hv) {
        const int q = lane >> 4;
        asm volatile("" ::: "memory");   // keep the weight fragments in LDS, not in registers (see gru_core.h)
        const double* tab = reinterpret_cast<const double*>(lds + L::OFF_TAB);
        const char* bhv = lds + L::OFF_BHV + (size_t)((sig_h + 1) * 3 + (sig_v + 1)) * L::SZ_B + (size_t)q * 32;
        const V2* wr = reinterpret_cast<const V2*>(lds + L::OFF_WR) + q * 2;
        const V2* av = reinterpret_cast<const V2*>(lds + L::OFF_A) + lane;
        auto hk = [&](int kk) -> double { return kk < KT ? hh[kk] : hv[kk - KT]; };
        // remainder units on the VALU: this lane's rows 4 kt + q of [Wh ; Wv] against its own fragment values
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        auto rem_part = [&](int k0, int k1) {
#pragma unroll
            for (int kk = k0; kk < k1; ++kk) {
                const V2 w01 = wr[kk * 8];
                s0 = __builtin_fma(hk(kk), w01[0], s0);
                s1 = __builtin_fma(hk(kk), w01[1], s1);
            }
            if (rem > 2) {
#pragma unroll
                for (int kk = k0; kk < k1; ++kk) {
                    const V2 w23 = wr[kk * 8 + 1];
                    s2 = __builtin_fma(hk(kk), w23[0], s2);
                    s3 = __builtin_fma(hk(kk), w23[1], s3);
                }
            }
        };
        V4 acc[NFULL];
#pragma unroll
        for (int t = 0; t < NFULL; ++t) acc[t] = *reinterpret_cast<const V4*>(bhv + t * 128);
        auto mfma_part = [&](int g0, int g1) {          // k-step pairs [g0, g1) of the concatenated [h_h ; h_v]
#pragma unroll
            for (int g = g0; g < g1; ++g) {
                V2 a[NFULL];
#pragma unroll
                for (int t = 0; t < NFULL; ++t) a[t] = av[(t * KT + g) * 64];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < NFULL; ++t) acc[t] = F::mfma(a[t][j], hk(2 * g + j), acc[t]);
            }
        };
        constexpr int GH = KT / 2;                      // pairs made of h_h k-steps only
        if (!RNNWF_ABLATED(ablate, 16)) rem_part(0, KT);
        if (!RNNWF_ABLATED(ablate, 32)) mfma_part(0, GH);
        fetch_hv();
        if (!RNNWF_ABLATED(ablate, 16)) rem_part(KT, 2 * KT);
        if (!RNNWF_ABLATED(ablate, 32)) mfma_part(GH, KT);
        // quarter q keeps unit j = q: lower half folds (s0, s1), upper half (s2, s3); then even rows keep the first
        const double ka = pair_sum32(s0, s2), kb = pair_sum32(s1, s3);
        double mine = pair_sum16(ka, kb);
        mine += *reinterpret_cast<const double*>(bhv + NFULL * 128);
        if (RNNWF_ABLATED(ablate, 8)) {                 // timing only: no elu
#pragma unroll
            for (int m = 0; m < NFULL; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[4 * m + r] = acc[m][r] * 0.001;
            out[KT - 1] = mine * 0.001;
            return;
        }
#pragma unroll
        for (int m = 0; m < NFULL; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[4 * m + r] = elu(acc[m][r], tab);
        out[KT - 1] = q < rem ? elu(mine, tab) : 0.0;
    }

    // log p(0), log p(1) of the Dense(2)+softmax head (and p(0) for the sampler), from the logit difference:
    // log p0 = -softplus(d), log p1 = -softplus(-d), d = z1 - z0; one exp and one log per site.
    static __device__ __forceinline__ void head(const char* lds, const double (&h)[KT], int lane, double& lp0,
                                                double& lp1, double& p0) {
        const int q = lane >> 4;
        asm volatile("" ::: "memory");
        const double* wdd = reinterpret_cast<const double*>(lds + L::OFF_WDD);
        double d = 0.0;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) d = __builtin_fma(h[kt], wdd[kt * 4 + q], d);
        d = quarter_sum(d) + wdd[KT * 4];
        const double* tab = reinterpret_cast<const double*>(lds + L::OFF_TAB);
        const double ad = __builtin_fabs(d);
        const double e = exp_tab(-ad, tab);                 // in (0, 1]
        const double lg = log1p_tab(e, tab);
        const double big = -(ad + lg), small = -lg;         // log-probability of the less / more likely value
        lp0 = d > 0.0 ? big : small;
        lp1 = d > 0.0 ? small : big;
        const double inv = rcp_fast_f64(1.0 + e);  // (sampler only)
        p0 = d > 0.0 ? e * inv : inv;
    }
};

__device__ __forceinline__ int md_spin(const uint32_t* bits, int64_t ns, int64_t s, int p) {
    return (int)((bits[(int64_t)(p >> 5) * ns + s] >> (p & 31)) & 1);
}

template <int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, NFULL <= 3 ? 2 : 1) mdrnn_base_kernel(MdArgs a) {
    using C = MdCore<NFULL>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, a.wimg);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int W = (N + 31) / 32;
    for (int64_t sb = gw; sb < a.nsb; sb += nw) {
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        double hh[KT], hv[KT];  // horizontal / vertical neighbour states
        double hn[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) hn[kt] = 0.0;
        uint32_t words[8];      // spins drawn / read so far (N <= 256)
#pragma unroll
        for (int w = 0; w < 8; ++w) words[w] = (!a.sampling && w < W) ? a.bits[(int64_t)w * a.ns + sc] : 0u;
        int sig_prev = -1;
        double cum = 0.0;
        for (int p = 0; p < N; ++p) {
            const int pv = a.vert_pos[p];
            const bool first = a.row_first[p] != 0;
            // horizontal neighbour = previous visit unless this site starts a row
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) hh[kt] = first ? 0.0 : hn[kt];
            const int sig_h = first ? -1 : sig_prev;
            int sig_v = -1;
            if (pv < 0) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hv[kt] = 0.0;
            } else if (pv == p - 1) {           // row turn: the vertical neighbour was computed one step ago
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hv[kt] = hn[kt];
                sig_v = sig_prev;
            } else {
                C::load_state(a.hs + (((int64_t)pv * a.nsb + sb) * C::KP) * 128 + 2 * lane, hv);
                uint32_t wv = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) if (w == (pv >> 5)) wv = words[w];
                sig_v = (wv >> (pv & 31)) & 1;
            }
            C::step(lds, sig_h, sig_v, hh, hv, hn, lane, a.rem);
            double lp0, lp1, p0;
            C::head(lds, hn, lane, lp0, lp1, p0);
            int sig;
            if (a.sampling) {
                const float u = philox_uniform(a.seed, a.step, (uint64_t)(a.sample_offset + sc), p);
                sig = ((double)u < p0) ? 0 : 1;
#pragma unroll
                for (int w = 0; w < 8; ++w) if (w == (p >> 5)) words[w] |= (uint32_t)sig << (p & 31);
            } else {
                uint32_t wp = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) if (w == (p >> 5)) wp = words[w];
                sig = (wp >> (p & 31)) & 1;
            }
            const double lsel = sig ? lp1 : lp0;
            if (a.lpq && valid && q == 0) a.lpq[(int64_t)a.row_of_pos[p] * a.ns + s] = cum + (sig ? lp0 : lp1);
            cum += lsel;
            if (a.hs) C::store_state(a.hs + (((int64_t)p * a.nsb + sb) * C::KP) * 128 + 2 * lane, hn);
            sig_prev = sig;
        }
        if (valid && q == 0) {
            if (a.sampling)
#pragma unroll
                for (int w = 0; w < 8; ++w) if (w < W) a.bits[(int64_t)w * a.ns + s] = words[w];
            if (a.lpq) a.lpq[s] = cum;
            if (a.out_lp) a.out_lp[s] = cum;
        }
    }
}

template <int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, NFULL <= 3 ? 2 : 1) mdrnn_flip_kernel(MdArgs a) {
    using C = MdCore<NFULL>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, a.wimg);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int W = (N + 31) / 32;
    // States this chain produced itself: ONE slot per lattice column.  On the zig-zag path row ny+1 consumes the states
    // of row ny in reverse order of their production, so the state above site (nx, ny+1) is always the last one written
    // to column nx, and the site's own state replaces it: Nx slots (84 KB per wave at 12 columns), half of what a
    // position-indexed ring of 2 Nx needs - 172 MB for the whole grid at config 4, which fits the 256 MB Infinity Cache.
    const int Nx = a.Nx;
    double* ring = a.ring + (int64_t)gw * Nx * C::KP * 128 + 2 * lane;
    uint32_t* words = reinterpret_cast<uint32_t*>(lds + C::L::BYTES) + (threadIdx.x >> 6) * 8 * 64 + lane;
    for (int64_t tile = gw; tile < a.ntiles; tile += nw) {
        const int i = (int)(tile / a.nsb);
        const int64_t sb = tile - (int64_t)i * a.nsb;
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        // the flipped configuration's spin words live in LDS ([word][lane], private to the wave): three 4-byte reads per
        // step at a wave-uniform word index instead of three 8-way register selects (48 VALU instructions per step)
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t v = w < W ? a.bits[(int64_t)w * a.ns + sc] : 0u;
            if (w == (i >> 5)) v ^= 1u << (i & 31);
            words[w * 64] = v;
        }
        double hv[KT], hn[KT];
        C::load_state(a.hs + (((int64_t)i * a.nsb + sb) * C::KP) * 128 + 2 * lane, hn);   // state after position i (unchanged)
        // h_v operand of position p: zero (first row), the state just computed (row turn), a base-pass state
        // (pv <= i) or one this chain produced (ring).  (Fetching one step ahead, behind the previous position's
        // head, measured 3 % SLOWER at config 4, and non-temporal ring accesses made no difference: the loads are
        // not latency-bound at 2 waves/SIMD.)
        // position bookkeeping of the zig-zag path without table look-ups: p = ny Nx + j, column nx = j (even rows) or
        // Nx-1-j (odd rows), vertical neighbour (nx, ny-1) at position p - 2j - 1
        int ny = (i + 1) / Nx, j = (i + 1) - ny * Nx;
        double lp = 0.0;
        // the three spin bits a step needs (horizontal input, vertical input, the observed spin) are looked up ONE step
        // ahead: the LDS reads, shifts and the table address of the accumulator start value leave the critical path
        auto bits_of = [&](int p, int jj, int nyy, int& sh, int& sv, int& so) {
            const int pv = nyy > 0 ? p - 2 * jj - 1 : -1;
            sh = jj == 0 ? -1 : (int)((words[((p - 1) >> 5) * 64] >> ((p - 1) & 31)) & 1);
            sv = pv >= 0 ? (int)((words[(pv >> 5) * 64] >> (pv & 31)) & 1) : -1;
            so = (int)((words[(p >> 5) * 64] >> (p & 31)) & 1);
        };
        int sig_h = -1, sig_v = -1, sig_o = 0;
        if (i + 1 < N) bits_of(i + 1, j, ny, sig_h, sig_v, sig_o);
        for (int p = i + 1; p < N; ++p) {
            const bool first = j == 0;
            const int pv = ny > 0 ? p - 2 * j - 1 : -1;
            const int nx = (ny & 1) ? Nx - 1 - j : j;
            // h_v operand: zero (first row), the state just computed (row turn), a base-pass state (pv <= i) or one this
            // chain produced (its column slot).  A row turn copies hn into hv before hn is cleared.
            if (pv < 0) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hv[kt] = 0.0;
            } else if (pv == p - 1) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hv[kt] = hn[kt];
            } else if (RNNWF_ABLATED(a.ablate, 2)) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hv[kt] = hn[kt] * 0.5;
            } else if (pv <= i) {
                C::load_state(a.hs + (((int64_t)pv * a.nsb + sb) * C::KP) * 128 + 2 * lane, hv);
            } else {
                C::load_state(ring + (int64_t)nx * C::KP * 128, hv);
            }
            if (first) {                           // no horizontal neighbour: uniform, once per row
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hn[kt] = 0.0;
            }
            const int sh = sig_h, sv = sig_v, so = sig_o;
            int jn = j + 1, nyn = ny;
            if (jn == Nx) { jn = 0; ++nyn; }
            if (p + 1 < N) bits_of(p + 1, jn, nyn, sig_h, sig_v, sig_o);      // next step's bits, in flight during this one
            C::step(lds, sh, sv, hn, hv, hn, lane, a.rem, a.ablate);
            double lp0 = hn[0], lp1 = hn[1], p0;
            if (!RNNWF_ABLATED(a.ablate, 4)) C::head(lds, hn, lane, lp0, lp1, p0);
            lp += so ? lp1 : lp0;
            // the last row has no vertical successor: nothing reads its states
            if (p < N - Nx && !RNNWF_ABLATED(a.ablate, 1)) C::store_state(ring + (int64_t)nx * C::KP * 128, hn);
            j = jn; ny = nyn;
        }
        if (valid && q == 0) a.lpq[(int64_t)a.row_of_pos[i] * a.ns + s] += lp;
    }
}

#ifdef RNNWF_DIAGNOSTICS
// The flip pass with the vertical neighbour's state PREFETCHED one step ahead by LDS-DMA (global_load_lds_dwordx4: global
// memory -> LDS without passing through registers; the kernel above has none to spare - 250 VGPRs at two waves per SIMD).
// Per wave one 7 KB staging slot in LDS: the transfer of step p + 1's h_v (a base-pass state or one of the chain's own column
// slots, written at least two steps earlier) starts in the middle of step p, right after step p's own h_v has been read out of
// the slot, and has the rest of step p plus the h_h half of step p + 1 (~5 000 cycles) to land; the kernel above issues the
// same 7 KB as register loads at the top of the step that needs them and waits ~1 000 cycles of their ~2 300 (profiles/
// r02_cfg4_ablation.md: 4 - 5 ms of 23.5).  One workgroup of 8 waves per CU (one weight image instead of two).  Same
// arithmetic, bit for bit.  MEASURED NEGATIVE (round 3, profiles/r03_d_cfg4_prefetch.md): 23.16 / 23.22 ms against 22.76 / 22.83 ms
// of the kernel above on one box, alternating - the wait it removes is not where the time is: the pass runs at 0.93 of what
// its f64 MFMA + f64 VALU instruction counts allow (they serialise on gfx950), and the slot's seven extra LDS reads and the
// DMA issue cost more than the hidden latency gains.  Kept behind RNNWF_MDRNN_PREFETCH=1 for A/B runs only.
template <int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 2) mdrnn_flip_pf_kernel(MdArgs a) {
    using C = MdCore<NFULL>;
    typedef typename C::V2 V2;
    constexpr int KT = C::KT, KP = C::KP;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, a.wimg);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t gw = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int W = (N + 31) / 32;
    const int Nx = a.Nx;
    double* ring = a.ring + (int64_t)gw * Nx * KP * 128 + 2 * lane;
    uint32_t* words = reinterpret_cast<uint32_t*>(lds + C::L::BYTES) + wave * 8 * 64 + lane;
    char* slot = lds + C::L::BYTES + (size_t)WAVES * C::L::WORDS_BYTES + (size_t)wave * KP * 1024;      // [KP][64] double2, this wave's
    typedef __attribute__((address_space(3))) void* LdsVoid;
    typedef const __attribute__((address_space(1))) void* GlobVoid;
    // global -> LDS: lane l's 16 bytes of piece g land at slot + g KB + 16 l (the layout load_state reads)
    auto dma = [&](const double* base) {           // base includes + 2 * lane
#pragma unroll
        for (int g = 0; g < KP; ++g)
            __builtin_amdgcn_global_load_lds((GlobVoid)(base + (size_t)g * 128), (LdsVoid)(slot + g * 1024), 16, 0, 0);
    };
    for (int64_t tile = gw; tile < a.ntiles; tile += nw) {
        const int i = (int)(tile / a.nsb);
        const int64_t sb = tile - (int64_t)i * a.nsb;
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t v = w < W ? a.bits[(int64_t)w * a.ns + sc] : 0u;
            if (w == (i >> 5)) v ^= 1u << (i & 31);
            words[w * 64] = v;
        }
        double hv[KT], hn[KT];
        C::load_state(a.hs + (((int64_t)i * a.nsb + sb) * KP) * 128 + 2 * lane, hn);
        int ny = (i + 1) / Nx, j = (i + 1) - ny * Nx;
        double lp = 0.0;
        auto bits_of = [&](int p, int jj, int nyy, int& sh, int& sv, int& so) {
            const int pv = nyy > 0 ? p - 2 * jj - 1 : -1;
            sh = jj == 0 ? -1 : (int)((words[((p - 1) >> 5) * 64] >> ((p - 1) & 31)) & 1);
            sv = pv >= 0 ? (int)((words[(pv >> 5) * 64] >> (pv & 31)) & 1) : -1;
            so = (int)((words[(p >> 5) * 64] >> (p & 31)) & 1);
        };
        // where position p's vertical neighbour comes from: 0 nothing (first row), 1 the state just computed (row turn),
        // 2 memory - a base-pass state (pv <= i) or the chain's own column slot
        auto source = [&](int p, int jj, int nyy, const double*& src) -> int {
            const int pv = nyy > 0 ? p - 2 * jj - 1 : -1;
            if (pv < 0) return 0;
            if (pv == p - 1) return 1;
            const int nx = (nyy & 1) ? Nx - 1 - jj : jj;
            src = pv <= i ? a.hs + (((int64_t)pv * a.nsb + sb) * KP) * 128 + 2 * lane : ring + (int64_t)nx * KP * 128;
            return 2;
        };
        int sig_h = -1, sig_v = -1, sig_o = 0;
        if (i + 1 < N) bits_of(i + 1, j, ny, sig_h, sig_v, sig_o);
        bool stores_behind = false;                    // ring stores issued after the transfer the coming step waits for
        {
            const double* src = nullptr;
            if (i + 1 < N && source(i + 1, j, ny, src) == 2) dma(src);
        }
        for (int p = i + 1; p < N; ++p) {
            const bool first = j == 0;
            const int nx = (ny & 1) ? Nx - 1 - j : j;
            const double* src_now = nullptr;
            const int kind = source(p, j, ny, src_now);
            if (kind == 0) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hv[kt] = 0.0;
            } else if (kind == 1) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hv[kt] = hn[kt];
            }
            if (first) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) hn[kt] = 0.0;
            }
            const int sh = sig_h, sv = sig_v, so = sig_o;
            int jn = j + 1, nyn = ny;
            if (jn == Nx) { jn = 0; ++nyn; }
            if (p + 1 < N) bits_of(p + 1, jn, nyn, sig_h, sig_v, sig_o);
            const double* src_next = nullptr;
            const int kind_next = p + 1 < N ? source(p + 1, jn, nyn, src_next) : 0;
            C::step_fetch(lds, sh, sv, hn, hv, hn, lane, a.rem, a.ablate, [&]() {
                if (kind == 2) {
                    // this step's h_v: its transfer was issued in the middle of the previous step (or at the tile's start); the only
                    // vector-memory operations behind it are that step's KP ring stores (counters: vmcnt 6 bits, lgkmcnt 4 bits)
                    if (stores_behind) __builtin_amdgcn_s_waitcnt(0x0F70 | (KP & 15) | ((KP >> 4) << 14));
                    else __builtin_amdgcn_s_waitcnt(0x0F70);
                    asm volatile("" ::: "memory");
                    const V2* st = reinterpret_cast<const V2*>(slot) + lane;
#pragma unroll
                    for (int g = 0; g < KP; ++g) {
                        const V2 v = st[g * 64];
                        hv[2 * g] = v[0];
                        if (2 * g + 1 < KT) hv[2 * g + 1] = v[1];
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): the slot has been read before it is refilled
                    asm volatile("" ::: "memory");
                }
                if (kind_next == 2) dma(src_next);
            });
            double lp0 = hn[0], lp1 = hn[1], p0;
            if (!RNNWF_ABLATED(a.ablate, 4)) C::head(lds, hn, lane, lp0, lp1, p0);
            lp += so ? lp1 : lp0;
            stores_behind = p < N - Nx;
            if (stores_behind) C::store_state(ring + (int64_t)nx * KP * 128, hn);
            j = jn; ny = nyn;
        }
        if (valid && q == 0) a.lpq[(int64_t)a.row_of_pos[i] * a.ns + s] += lp;
    }
}

#endif  // RNNWF_DIAGNOSTICS

}  // namespace rnnwf
