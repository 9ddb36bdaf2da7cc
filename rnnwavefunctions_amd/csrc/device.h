// device.h - gfx950 device helpers: MFMA fragment traits, transcendental helpers, Philox4x32-10.
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"

namespace rnnwf {

// ---- 16x16x4 MFMA, f32 and f64 -------------------------------------------------------------------
// A: lane l holds A[row = l & 15][k = l >> 4];  B: lane l holds B[k = l >> 4][col = l & 15].
// C/D f32: lane l, reg r -> D[4 (l >> 4) + r][l & 15];  f64: D[(l >> 4) + 4 r][l & 15].
template <typename T> struct Frag;
template <> struct Frag<float> {
    typedef float V4 __attribute__((ext_vector_type(4)));
    typedef float VA __attribute__((ext_vector_type(4)));
    static constexpr int VW = 4;
    static __device__ __forceinline__ V4 mfma(float a, float b, V4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};
template <> struct Frag<double> {
    typedef double V4 __attribute__((ext_vector_type(4)));
    typedef double VA __attribute__((ext_vector_type(2)));
    static constexpr int VW = 2;
    static __device__ __forceinline__ V4 mfma(double a, double b, V4 c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
};

// ---- activations -----------------------------------------------------------------------------------
// f32: v_exp_f32 / v_rcp_f32 (1 ulp each); abs error of sigmoid/tanh ~1e-7, the f32 noise floor of
// the recurrence.  f64: ocml.
__device__ __forceinline__ float sigmoid_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float tanh_(float x) {
    const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * x);  // exp(-2x)
    return (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ double sigmoid_(double x) { return 1.0 / (1.0 + exp(-x)); }
__device__ __forceinline__ double tanh_(double x) { return tanh(x); }
// exp for x in [-700, 700] without the special-case handling of ocml: k = rint(x log2 e), r = x - k ln2 in two
// steps (fdlibm's ln2 split), degree-13 Taylor polynomial on |r| <= 0.3466 (truncation 4e-18), v_ldexp_f64.
// 20 VALU instructions against ~40 for ocml's exp and ~60 (with branches) for expm1; relative error < 3e-16.
__device__ __forceinline__ double exp_fast(double x) {
    x = x < -700.0 ? -700.0 : x;
    const double kf = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(kf, -6.93147180369123816490e-01, x);
    r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}
// ---- table-driven f64 exp / log for the per-site elementwise work of the float64 models -------------------------------
// On gfx950 the f64 MFMA and the f64 VALU do not overlap at all - neither inside a wave nor across the waves of a SIMD
// (tools/microbench/issue_model: k_roles_g, g_q*) - so every VALU instruction of the MDRNN step is exposed time and the
// 13 exp() of a 16-chain wave-step are worth shortening.  A 3 x 64-entry table in LDS (F64Tables, 1.5 KB, written
// by the host packers) cuts exp from 21 to 16 instructions and log(1 + e), e in (0, 1], from ocml's ~60 to 14.
// 1/d to ~1 ulp: v_rcp_f64 plus two Newton steps (5 instructions instead of an IEEE division's ~15)
__device__ __forceinline__ double rcp_fast_f64(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(y, e, y);
}
// ---- cross-quarter exchanges of doubles on v_permlane16_swap / v_permlane32_swap (VALU, 4 cycles per dword) instead of
// ds_bpermute (an LDS round trip each): swap32(a, b) leaves {a.lower | b.lower} and {a.upper | b.upper} (lane halves),
// swap16 the same for the 16-lane rows {r0, r2} / {r1, r3}; the sum of the two results is, per half (row pair), the
// reduction of a over the halves in the lower one and of b in the upper one.
__device__ __forceinline__ double pair_sum32(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double pair_sum16(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// sum of v over the four lane quarters, in every lane
__device__ __forceinline__ double quarter_sum(double v) { const double t = pair_sum16(v, v); return pair_sum32(t, t); }

// e^x for x <= 0: x = (64 m + j) ln2/64 + r, |r| <= ln2/128; e^r by a degree-5 polynomial (truncation r^6/720 < 4e-17);
// relative error ~1 ulp of the tabulated 2^(j/64).  No clamp: far below the underflow threshold the integer conversion
// saturates, r stays finite and v_ldexp_f64 returns 0.
__device__ __forceinline__ double exp_tab(double x, const double* tab) {
    const double kf = __builtin_rint(x * 0x1.71547652b82fep+6);                  // 64 / ln 2
    double r = __builtin_fma(kf, -0x1.62e42fee00000p-7, x);                     // ln2/64, high part (21 trailing zero bits: k * high is exact)
    r = __builtin_fma(kf, -0x1.a39ef35793c76p-39, r);                            // ln2/64 - high
    const int k = (int)kf;
    const double t = tab[F64Tables::EXP2 + (k & 63)];
    double q = __builtin_fma(r, 1.0 / 120.0, 1.0 / 24.0);
    q = __builtin_fma(q, r, 1.0 / 6.0);
    q = __builtin_fma(q, r, 0.5);
    q = __builtin_fma(q, r, 1.0);
    return ldexp(__builtin_fma(t, r * q, t), k >> 6);
}
// log(1 + e) for e in [0, 1]: y = 1 + e = c_j (1 + u), |u| <= 2^-7; log1p(u) to degree 7 (truncation u^8/8 < 3e-18)
__device__ __forceinline__ double log1p_tab(double e, const double* tab) {
    const double y = 1.0 + e;
    int j = (int)__builtin_fma(y, 64.0, -64.0);
    j = j > 63 ? 63 : j;
    const double u = __builtin_fma(y, tab[F64Tables::RCPC + j], -1.0);
    double p = __builtin_fma(u, 1.0 / 7.0, -1.0 / 6.0);
    p = __builtin_fma(p, u, 1.0 / 5.0);
    p = __builtin_fma(p, u, -0.25);
    p = __builtin_fma(p, u, 1.0 / 3.0);
    p = __builtin_fma(p, u, -0.5);
    p = __builtin_fma(p, u, 1.0);
    return __builtin_fma(p, u, tab[F64Tables::LOGC + j]);
}
__device__ __forceinline__ float exp_(float x) { return expf(x); }
__device__ __forceinline__ double exp_(double x) { return exp(x); }
__device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }

// ---- Philox4x32-10 (oracle/philox.py is the NumPy twin) --------------------------------------------
struct Philox4 { uint32_t v[4]; };
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 r; r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
    return r;
}
// uniform in [0,1), a multiple of 2^-24, for (global sample g, site n) of VMC step `step`
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t step, uint64_t g, int n) {
    const Philox4 r = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(n >> 2), (uint32_t)step,
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
    const int j = n & 3;
    const uint32_t w = j == 0 ? r.v[0] : j == 1 ? r.v[1] : j == 2 ? r.v[2] : r.v[3];
    return (float)(w >> 8) * 5.9604644775390625e-08f;
}

}  // namespace rnnwf
