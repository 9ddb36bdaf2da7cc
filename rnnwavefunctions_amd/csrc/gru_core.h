// gru_core.h - one recurrent step of the cuDNN-compatible GRU for 16 chains per wave, on MFMA.
//
// Reference arithmetic: tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell.call as invoked at
// 1DTFIM/RNNwavefunction.py:66,108 and J1J2/ComplexRNNwavefunction.py:80,140:
//     g = sigmoid([x,h] Wg + bg);  r,u = split(g)
//     c = tanh(x Wci + bci + r * (h Wch + bch));   h' = (1-u) c + u h
// x is a one-hot (or the zero vector at the first site), so its matmuls are row selections that are
// folded into the accumulator initialisation (tables BINIT / XC of layout.h).
//
// f32 activations run on v_exp_f32 (2^x): the packed image carries the gate rows pre-multiplied by
// -log2(e) and the candidate rows by -2 log2(e) (pack.h, Act<float>), so that
//     sigmoid(x) = 1 / (1 + 2^acc)          tanh(y) = 2 / (1 + 2^pre) - 1
// need no multiply; h' = c + u (h - c).  f64 keeps unscaled rows and libm-grade exp/tanh.
#pragma once
#include "device.h"

namespace rnnwf {

// Kernels that must agree bit for bit (prnn_base_kernel / prnn_base_coop_kernel, crnn_* likewise) share the functions
// below.  -ffp-contract=fast lets hipcc fuse a*b+c differently in every inlining context, so these functions switch
// contraction off and spell their FMAs out: the same source then gives the same rounding everywhere.
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T> struct Act;
template <> struct Act<float> {
    static constexpr double kGateScale = -1.44269504088896340736;      // acc = -x log2(e)
    static constexpr double kCandScale = -2.88539008177792681472;      // pre = -2 y log2(e)
    static __device__ __forceinline__ float sigmoid_scaled(float a) {
#pragma clang fp contract(off)
        return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
    }
    static __device__ __forceinline__ float tanh_scaled(float p) {
#pragma clang fp contract(off)
        return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p)), -1.0f);
    }
};
template <> struct Act<double> {
    static constexpr double kGateScale = 1.0;
    static constexpr double kCandScale = 1.0;
    // f64 without ocml's special-case handling: exp_fast (20 instructions, device.h) and a reciprocal from v_rcp_f64
    // plus two Newton steps (5 instructions) instead of exp (~40) + IEEE division (~15) and tanh (~60 with branches);
    // relative error < 1e-15, i.e. the level of the f64 oracle's own libm.  2DTFIM_1DRNN flip pass 58 -> see DESIGN.md.
    static __device__ __forceinline__ double rcp_fast(double d) { return rcp_fast_f64(d); }
    static __device__ __forceinline__ double sigmoid_scaled(double a) {
        const double x = a > 700.0 ? 700.0 : a;                    // exp_fast clamps the other side
        return rcp_fast(1.0 + exp_fast(-x));
    }
    static __device__ __forceinline__ double tanh_scaled(double p) {
        return __builtin_fma(2.0, sigmoid_scaled(2.0 * p), -1.0);    // absolute error ~1e-16 (the state update adds, never divides)
    }
};

// log-probabilities of a two-way softmax from the logit difference d = z1 - z0:
//   log p1 = -log(1 + e^-d), log p0 = -log(1 + e^d); evaluated as  -[max(0, -+d)] - log(1 + e^-|d|)
// (tf.nn.softmax's exp(z - max)/sum followed by log, 1DTFIM/RNNwavefunction.py:109,113-116, in one step).
__device__ __forceinline__ void log_softmax2(float d, float& lp0, float& lp1) {
#pragma clang fp contract(off)
    const float ad = fabsf(d);
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * ad);
    const float L = 0.693147180559945309f * __builtin_amdgcn_logf(1.0f + e);     // v_log_f32 = log2
    lp0 = -L - fmaxf(d, 0.0f);
    lp1 = -L - fmaxf(-d, 0.0f);
}
__device__ __forceinline__ void log_softmax2(double d, double& lp0, double& lp1) {
#pragma clang fp contract(off)
    const double L = log1p(exp(-fabs(d)));
    lp0 = -L - fmax(d, 0.0);
    lp1 = -L - fmax(-d, 0.0);
}
// p0 = sigmoid(-d) for the sampler (tf.multinomial draws class 0 iff u * (p0 + p1) < p0)
__device__ __forceinline__ float prob0(float d) {
#pragma clang fp contract(off)
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(1.44269504088896341f * d));
}
__device__ __forceinline__ double prob0(double d) { return 1.0 / (1.0 + exp(d)); }

// new state of one unit from its three accumulators (pre-scaled rows, see Act), the candidate's input term and the
// old state:  r = sigmoid(a_r), u = sigmoid(a_u), c = tanh(xc + r q), h' = c + u (h - c)
template <typename T>
__device__ __forceinline__ T gru_gate(T ar, T au, T aq, T xc, T hold) {
#pragma clang fp contract(off)
    const T rg = Act<T>::sigmoid_scaled(ar);
    const T ug = Act<T>::sigmoid_scaled(au);
    const T cc = Act<T>::tanh_scaled(fma_(rg, aq, xc));
    return fma_(ug, hold - cc, cc);
}

template <typename T, int NFULL, int NOUT>
struct GruCore {
    using L = GruLayout<T, NFULL, NOUT>;
    using F = Frag<T>;
    using A = Act<T>;
    using V4 = typename F::V4;
    using VA = typename F::VA;
    static constexpr int KT = L::KT, NT = L::NT, NG = L::NG, VW = L::VW;
    static constexpr int TC = 5;  // tiles per MFMA issue group (independent accumulators back to back)

    // Stage the packed weight image into LDS (all threads of the workgroup) and return where the step functions find it: LDS, or
    // - L::SPILL, the image exceeds LDS - the global image itself.
    static __device__ __forceinline__ const char* stage(char* lds, const void* wimg) {
        if constexpr (L::SPILL) {
            return reinterpret_cast<const char*>(wimg);
        } else {
            const uint4* src = reinterpret_cast<const uint4*>(wimg);
            uint4* dst = reinterpret_cast<uint4*>(lds);
            for (int i = threadIdx.x; i < (int)(L::BYTES / 16); i += blockDim.x) dst[i] = src[i];
            __syncthreads();
            return lds;
        }
    }
    // L::SPILL: the products  acc += A h  with the fragments read through L2 - buffer loads with the fragment's position as the scalar
    // offset, two groups of five tiles ahead of their MFMAs and no further (UpperCore::block explains why)
    static __device__ __forceinline__ void mfma_streamed(const char* img, const T (&h)[KT], V4 (&acc)[NT], int lane) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img), 0, (int)L::OFF_BINIT, 0x00020000);
        // (133..260 units run one wave per SIMD - nothing else hides the L2 latency: four groups ahead instead of two)
        constexpr int NTG = (NT + TC - 1) / TC, NGR = NG * NTG, AHEAD = NFULL >= 12 ? 4 : 2;
        VA ring[AHEAD + 1][TC];
        auto request = [&](int k, VA (&dstv)[TC]) {
            const int g = k / NTG, t0 = (k % NTG) * TC;
#pragma unroll
            for (int t = 0; t < TC; ++t)
                if (t0 + t < NT) {
                    const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(rv, lane * 16, ((t0 + t) * NG + g) * 64 * 16, 0);
                    dstv[t] = __builtin_bit_cast(VA, w);
                }
        };
#pragma unroll
        for (int k = 0; k < AHEAD && k < NGR; ++k) request(k, ring[k]);
#pragma unroll
        for (int k = 0; k < NGR; ++k) {
            if (k + AHEAD < NGR) request(k + AHEAD, ring[(k + AHEAD) % (AHEAD + 1)]);
            asm volatile("" ::: "memory");
            // one wave per SIMD: left alone, the scheduler sinks every request down to its first use (short live ranges) and the
            // wave waits out a full L2 round trip per group - 60 % of its cycles at 260 units (profiles/r04_m_wide_widths.txt)
            if constexpr (NFULL >= 12) __builtin_amdgcn_sched_barrier(0);
            const int g = k / NTG, t0 = (k % NTG) * TC;
#pragma unroll
            for (int j = 0; j < VW; ++j)
#pragma unroll
                for (int t = 0; t < TC; ++t)
                    if (t0 + t < NT) acc[t0 + t] = F::mfma(ring[k % (AHEAD + 1)][t][j], h[g * VW + j], acc[t0 + t]);
            if constexpr (NFULL >= 12) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            T a;
            if constexpr (sizeof(T) == 4) {
                a = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rv, lane * 4, (int)L::OFF_AREM + t * 64 * 4, 0));
            } else {
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rv, lane * 8, (int)L::OFF_AREM + t * 64 * 8, 0);
                a = __builtin_bit_cast(T, w);
            }
            acc[t] = F::mfma(a, h[KT - 1], acc[t]);
        }
    }

    // h[kt] of lane (c, q) holds unit 4 kt + q of chain c.  sig: input spin of this step (-1: zero vector).
    // `ablate` (diagnostics, RNNWF_ABLATE): 1 skips the MFMAs, 2 the gate arithmetic - used to measure that on
    // gfx950 f32 MFMA time and VALU time ADD (no overlap across the waves of a SIMD): see DESIGN.md section 6.
    // BIAS_LAST: the bias / one-hot-input rows are added AFTER the products (base pass: its cooperative variant,
    // prnn_base_coop_kernel, starts the products before the input spin is known and must agree bit for bit).
    template <bool BIAS_LAST = false>
    static __device__ __forceinline__ void step(const char* lds, int sig, T (&h)[KT], int lane, int ablate = 0) {
        step_impl(lds, sig, h, lane, ablate, BIAS_LAST);
    }
    // non-template entry for the stacked-layer cores (hipcc's host pass fails to resolve step<> from inside a second
    // class template for T = double)
    static __device__ __forceinline__ void step_plain(const char* lds, int sig, T (&h)[KT], int lane) {
        step_impl(lds, sig, h, lane, 0, false);
    }
    // BIAS_LAST is a compile-time constant at every call site (forced inlining folds the branches)
    static __device__ __forceinline__ void step_impl(const char* lds, int sig, T (&h)[KT], int lane, int ablate, const bool BIAS_LAST) {
        const int q = lane >> 4;
        // The weight image never changes, so the compiler would hoist all ~NT*KT fragment loads out of
        // the site loop and pin them in registers (1 wave/SIMD).  Re-read them from LDS every step.
        asm volatile("" ::: "memory");
        V4 acc[NT];
        const char* binit = lds + L::OFF_BINIT + (size_t)(sig + 1) * L::SZ_BINIT_VARIANT + (size_t)q * 4 * sizeof(T);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = BIAS_LAST ? V4{T(0), T(0), T(0), T(0)} : *reinterpret_cast<const V4*>(binit + (size_t)t * 16 * sizeof(T));
        const VA* av = reinterpret_cast<const VA*>(lds + L::OFF_AVEC) + lane;
        // The MFMA chain and the gate arithmetic stay separate scheduling regions: merged into one, hipcc weaves the
        // gates between the f32 MFMAs - which cannot overlap them on gfx950 - and pays 17 more hazard no-ops and 7 % of
        // config 5 (measured round 2: 320.8 vs 297.3 ms; round 1 had the regions by accident, through the run-time
        // diagnostics branches that stood here).
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (L::SPILL) {
            mfma_streamed(lds, h, acc, lane);
        } else if (!RNNWF_ABLATED(ablate, 1)) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int t0 = 0; t0 < NT; t0 += TC) {
                VA a[TC];
#pragma unroll
                for (int t = 0; t < TC; ++t)
                    if (t0 + t < NT) a[t] = av[((t0 + t) * NG + g) * 64];
#pragma unroll
                for (int j = 0; j < VW; ++j)
#pragma unroll
                    for (int t = 0; t < TC; ++t)
                        if (t0 + t < NT) acc[t0 + t] = F::mfma(a[t][j], h[g * VW + j], acc[t0 + t]);
            }
        }
        {
            const T* ar = reinterpret_cast<const T*>(lds + L::OFF_AREM) + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = F::mfma(ar[t * 64], h[KT - 1], acc[t]);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (BIAS_LAST) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] += *reinterpret_cast<const V4*>(binit + (size_t)t * 16 * sizeof(T));
        }
        if (RNNWF_ABLATED(ablate, 2)) {      // keep the accumulators alive without the gate arithmetic
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" :: "v"(acc[t]));
            return;
        }
        const char* x = lds + L::OFF_XC + (size_t)(sig + 1) * L::SZ_XC_VARIANT + (size_t)q * 4 * sizeof(T);
#pragma unroll
        for (int m = 0; m < NFULL; ++m) {
            const V4 xc = *reinterpret_cast<const V4*>(x + (size_t)m * 16 * sizeof(T));
#pragma unroll
            for (int r = 0; r < 4; ++r)
                h[4 * m + r] = gru_gate<T>(acc[m][r], acc[NFULL + m][r], acc[2 * NFULL + m][r], xc[r], h[4 * m + r]);
        }
        {
            const V4 xc = *reinterpret_cast<const V4*>(x + (size_t)NFULL * 16 * sizeof(T));
            const V4 a = acc[NT - 1];
            h[KT - 1] = gru_gate<T>(a[0], a[1], a[2], xc[0], h[KT - 1]);
        }
    }

    // Forward step that also returns the gate values of this lane's units (index kt <-> unit 4 kt + q), for the
    // backward pass: r, u, c and the candidate's hidden projection q = h Wch + bch (un-scaled).
    static __device__ __forceinline__ void step_keep(const char* lds, int sig, const T (&h)[KT], T (&hn)[KT], T (&rg)[KT],
                                                     T (&ug)[KT], T (&cc)[KT], T (&qv)[KT], int lane) {
        const int q = lane >> 4;
        asm volatile("" ::: "memory");
        V4 acc[NT];
        {
            const char* b = lds + L::OFF_BINIT + (size_t)(sig + 1) * L::SZ_BINIT_VARIANT + (size_t)q * 4 * sizeof(T);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = *reinterpret_cast<const V4*>(b + (size_t)t * 16 * sizeof(T));
        }
        if constexpr (L::SPILL) {
            mfma_streamed(lds, h, acc, lane);
        } else {
        const VA* av = reinterpret_cast<const VA*>(lds + L::OFF_AVEC) + lane;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int t0 = 0; t0 < NT; t0 += TC) {
                VA a[TC];
#pragma unroll
                for (int t = 0; t < TC; ++t)
                    if (t0 + t < NT) a[t] = av[((t0 + t) * NG + g) * 64];
#pragma unroll
                for (int j = 0; j < VW; ++j)
#pragma unroll
                    for (int t = 0; t < TC; ++t)
                        if (t0 + t < NT) acc[t0 + t] = F::mfma(a[t][j], h[g * VW + j], acc[t0 + t]);
            }
        }
        {
            const T* ar = reinterpret_cast<const T*>(lds + L::OFF_AREM) + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = F::mfma(ar[t * 64], h[KT - 1], acc[t]);
        }
        }
        const char* x = lds + L::OFF_XC + (size_t)(sig + 1) * L::SZ_XC_VARIANT + (size_t)q * 4 * sizeof(T);
        const T inv_cs = T(1.0 / A::kCandScale);
#pragma unroll
        for (int m = 0; m < NFULL; ++m) {
            const V4 xc = *reinterpret_cast<const V4*>(x + (size_t)m * 16 * sizeof(T));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * m + r;
                rg[k] = A::sigmoid_scaled(acc[m][r]);
                ug[k] = A::sigmoid_scaled(acc[NFULL + m][r]);
                qv[k] = acc[2 * NFULL + m][r] * inv_cs;
                cc[k] = A::tanh_scaled(xc[r] + rg[k] * acc[2 * NFULL + m][r]);
                hn[k] = cc[k] + ug[k] * (h[k] - cc[k]);
            }
        }
        {
            const V4 xc = *reinterpret_cast<const V4*>(x + (size_t)NFULL * 16 * sizeof(T));
            const V4 a = acc[NT - 1];
            rg[KT - 1] = A::sigmoid_scaled(a[0]);
            ug[KT - 1] = A::sigmoid_scaled(a[1]);
            qv[KT - 1] = a[2] * inv_cs;
            cc[KT - 1] = A::tanh_scaled(xc[0] + rg[KT - 1] * a[2]);
            hn[KT - 1] = cc[KT - 1] + ug[KT - 1] * (h[KT - 1] - cc[KT - 1]);
        }
    }

    // Output heads on the new hidden state, reduced over the four lane quarters.  Row 0 is the softmax
    // logit DIFFERENCE d = z1 - z0 of tf.layers.Dense(2) (1DTFIM/RNNwavefunction.py:33,67,109) - a two-way
    // softmax depends on nothing else; the cRNN adds the two phase logits (rows 1, 2; :42-43).
    static __device__ __forceinline__ void head(const char* lds, const T (&h)[KT], int lane, T (&z)[NOUT]) {
#pragma clang fp contract(off)
        const int q = lane >> 4;
        asm volatile("" ::: "memory");
        const T* wd = reinterpret_cast<const T*>(lds + L::OFF_WD) + q * L::WD_Q;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) z[o] = T(0);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) z[o] = fma_(h[kt], wd[kt * NOUT + o], z[o]);
        const T* bd = reinterpret_cast<const T*>(lds + L::OFF_BD);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            z[o] += __shfl_xor(z[o], 16);
            z[o] += __shfl_xor(z[o], 32);
            z[o] += bd[o];
        }
    }
};

// One step of a stacked GRU layer above the first (UpperLayout): x = new state of the layer below.
template <int NFULL, typename T = float>
struct UpperCore {
    using U = UpperLayout<NFULL, T>;
    using F = Frag<T>;
    using A = Act<T>;
    using V4 = typename F::V4;
    using VA = typename F::VA;
    static constexpr int KT = U::KT, NT = U::NT, NT2 = U::NT2, NG = U::NG, VW = U::VW;

    // `streamed` (a compile-time constant at every call site: forced inlining folds the branch): the image is in global memory
    // (layout.h: MlSpill), its fragments are requested two groups of five tiles ahead of their MFMAs - and no further: left alone,
    // hipcc issues every load of the block up front and spills hundreds of registers.
    template <bool XBLOCK>
    static __device__ __forceinline__ void block(const char* lds, const T (&v)[KT], V4 (&acc)[NT2], int lane, const bool streamed = false) {
        const VA* av = reinterpret_cast<const VA*>(lds + (XBLOCK ? U::OFF_AX : U::OFF_AH)) + lane;
        const T* ar = reinterpret_cast<const T*>(lds + (XBLOCK ? U::OFF_AXR : U::OFF_AHR)) + lane;
        // block tile t -> accumulator tile: r, u unchanged; third group -> y (X block) or q (H block); mixed last
        auto dst = [](int t) { return t < 2 * NFULL ? t : t < 3 * NFULL ? (XBLOCK ? t + NFULL : t) : NT2 - 1; };
        constexpr int TC = 5;
        if (streamed) {
            // buffer loads: one resource descriptor for the block's fragments, the lane's slot as the VGPR offset, the fragment's
            // position as the scalar offset - with flat loads hipcc keeps one loop-invariant 64-bit address per 4 KB window
            // (hundreds of registers at 100 units) and spills them
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(lds + (XBLOCK ? U::OFF_AX : U::OFF_AH)), 0, (int)(U::SZ_AVEC + U::SZ_AREM), 0x00020000);
            constexpr int NTG = (NT + TC - 1) / TC, NGR = NG * NTG, AHEAD = 2;
            VA ring[AHEAD + 1][TC];
            auto request = [&](int k, VA (&dstv)[TC]) {
                const int g = k / NTG, t0 = (k % NTG) * TC;
#pragma unroll
                for (int t = 0; t < TC; ++t)
                    if (t0 + t < NT) {
                        const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(rv, lane * 16, ((t0 + t) * NG + g) * 64 * 16, 0);
                        dstv[t] = __builtin_bit_cast(VA, w);
                    }
            };
#pragma unroll
            for (int k = 0; k < AHEAD && k < NGR; ++k) request(k, ring[k]);
#pragma unroll
            for (int k = 0; k < NGR; ++k) {
                if (k + AHEAD < NGR) request(k + AHEAD, ring[(k + AHEAD) % (AHEAD + 1)]);
                asm volatile("" ::: "memory");
                const int g = k / NTG, t0 = (k % NTG) * TC;
#pragma unroll
                for (int j = 0; j < VW; ++j)
#pragma unroll
                    for (int t = 0; t < TC; ++t)
                        if (t0 + t < NT) acc[dst(t0 + t)] = F::mfma(ring[k % (AHEAD + 1)][t][j], v[g * VW + j], acc[dst(t0 + t)]);
            }
            // the kt = KT - 1 column: [NT][64] T right behind the vectors (OFF_AXR = OFF_AX + SZ_AVEC, likewise for the H block)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                T a;
                if constexpr (sizeof(T) == 4) {
                    a = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rv, lane * 4, (int)U::SZ_AVEC + t * 64 * 4, 0));
                } else {
                    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                    const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rv, lane * 8, (int)U::SZ_AVEC + t * 64 * 8, 0);
                    a = __builtin_bit_cast(T, w);
                }
                acc[dst(t)] = F::mfma(a, v[KT - 1], acc[dst(t)]);
            }
            return;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int t0 = 0; t0 < NT; t0 += TC) {
                VA a[TC];
#pragma unroll
                for (int t = 0; t < TC; ++t)
                    if (t0 + t < NT) a[t] = av[((t0 + t) * NG + g) * 64];
#pragma unroll
                for (int j = 0; j < VW; ++j)
#pragma unroll
                    for (int t = 0; t < TC; ++t)
                        if (t0 + t < NT) acc[dst(t0 + t)] = F::mfma(a[t][j], v[g * VW + j], acc[dst(t0 + t)]);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[dst(t)] = F::mfma(ar[t * 64], v[KT - 1], acc[dst(t)]);
    }

    static __device__ __forceinline__ void products(const char* lds, const T (&x)[KT], const T (&h)[KT], V4 (&acc)[NT2], int lane,
                                                    const bool streamed = false) {
        const int q = lane >> 4;
        asm volatile("" ::: "memory");
        {
            const char* b = lds + U::OFF_B + (size_t)q * 4 * sizeof(T);
#pragma unroll
            for (int t = 0; t < NT2; ++t) acc[t] = *reinterpret_cast<const V4*>(b + (size_t)t * 16 * sizeof(T));
        }
        block<true>(lds, x, acc, lane, streamed);
        asm volatile("" ::: "memory");
        block<false>(lds, h, acc, lane, streamed);
    }

    // forward step that keeps this lane's gate values for the backward pass (q = h Wch + bch un-scaled)
    static __device__ __forceinline__ void step_keep(const char* lds, const T (&x)[KT], const T (&h)[KT],
                                                     T (&hn)[KT], T (&rg)[KT], T (&ug)[KT], T (&cc)[KT],
                                                     T (&qv)[KT], int lane, const bool streamed = false) {
        V4 acc[NT2];
        products(lds, x, h, acc, lane, streamed);
        const T inv_cs = (T)(1.0 / A::kCandScale);
#pragma unroll
        for (int m = 0; m < NFULL; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * m + r;
                rg[k] = A::sigmoid_scaled(acc[m][r]);
                ug[k] = A::sigmoid_scaled(acc[NFULL + m][r]);
                qv[k] = acc[2 * NFULL + m][r] * inv_cs;
                cc[k] = A::tanh_scaled(acc[3 * NFULL + m][r] + rg[k] * acc[2 * NFULL + m][r]);
                hn[k] = cc[k] + ug[k] * (h[k] - cc[k]);
            }
        {
            const V4 a = acc[NT2 - 1];
            rg[KT - 1] = A::sigmoid_scaled(a[0]);
            ug[KT - 1] = A::sigmoid_scaled(a[1]);
            qv[KT - 1] = a[2] * inv_cs;
            cc[KT - 1] = A::tanh_scaled(a[3] + rg[KT - 1] * a[2]);
            hn[KT - 1] = cc[KT - 1] + ug[KT - 1] * (h[KT - 1] - cc[KT - 1]);
        }
    }

    static __device__ __forceinline__ void step(const char* lds, const T (&x)[KT], T (&h)[KT], int lane, const bool streamed = false) {
        V4 acc[NT2];
        products(lds, x, h, acc, lane, streamed);
#pragma unroll
        for (int m = 0; m < NFULL; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T rg = A::sigmoid_scaled(acc[m][r]);
                const T ug = A::sigmoid_scaled(acc[NFULL + m][r]);
                const T cc = A::tanh_scaled(acc[3 * NFULL + m][r] + rg * acc[2 * NFULL + m][r]);
                h[4 * m + r] = cc + ug * (h[4 * m + r] - cc);
            }
        {
            const V4 a = acc[NT2 - 1];
            const T rg = A::sigmoid_scaled(a[0]);
            const T ug = A::sigmoid_scaled(a[1]);
            const T cc = A::tanh_scaled(a[3] + rg * a[2]);
            h[KT - 1] = cc + ug * (h[KT - 1] - cc);
        }
    }
};

}  // namespace rnnwf
