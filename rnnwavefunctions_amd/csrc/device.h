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
