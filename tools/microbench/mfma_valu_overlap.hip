// Micro-benchmark: does VALU/transcendental work of one wave overlap with MFMA work of ANOTHER wave on the same
// SIMD?  512-thread blocks (8 waves; waves w and w+4 share a SIMD), one block per CU.
//   mode 0: all 8 waves MFMA          mode 1: all 8 waves VALU
//   mode 2: waves 0-3 MFMA, 4-7 VALU  (same per-wave work as modes 0/1)
// for the f32 16x16x4 MFMA and the bf16 32x32x16 MFMA.  If the pipes are independent, t(2) ~ max(t(0), t(1))/1
// with per-wave work unchanged -> t(2) ~ max(t0, t1) / 2 ... reported raw.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short bf8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ float do_mfma(int iters, float seed) {
    if constexpr (KIND == 0) {
        f4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = seed, y = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
        return a0[0] + a1[1] + a2[2] + a3[3];
    } else if constexpr (KIND == 2) {
        float x = seed, y = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 a[0:3], %0, %1, a[0:3]\n\t"
                         "v_mfma_f32_16x16x4_f32 a[4:7], %0, %1, a[4:7]\n\t"
                         "v_mfma_f32_16x16x4_f32 a[8:11], %0, %1, a[8:11]\n\t"
                         "v_mfma_f32_16x16x4_f32 a[12:15], %0, %1, a[12:15]"
                         :: "v"(x), "v"(y) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
        }
        return x;
    } else if constexpr (KIND == 3) {
        bf8 x = {1, 2, 3, 4, 5, 6, 7, (short)seed}, y = {8, 7, 6, 5, 4, 3, 2, 1};
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n\t"
                         "v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]"
                         :: "v"(x), "v"(y) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15",
                            "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
        }
        return (float)x[0];
    } else {
        f16v a0 = {}, a1 = {};
        bf8 x = {1, 2, 3, 4, 5, 6, 7, (short)seed}, y = {8, 7, 6, 5, 4, 3, 2, 1};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        }
        return a0[0] + a1[5];
    }
}
__device__ __forceinline__ float do_fma(int iters, float seed) {     // plain (non-transcendental) VALU: 16 fma per iteration
    float v0 = seed, v1 = seed + 1.f, v2 = seed + 2.f, v3 = seed + 3.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v0 = fmaf(v0, 0.999f, 0.25f); v1 = fmaf(v1, 0.999f, 0.25f);
            v2 = fmaf(v2, 0.999f, 0.25f); v3 = fmaf(v3, 0.999f, 0.25f);
        }
    }
    return v0 + v1 + v2 + v3;
}
__device__ __forceinline__ float do_valu(int iters, float seed) {
    float v0 = seed, v1 = seed + 1.f, v2 = seed + 2.f, v3 = seed + 3.f;
    for (int i = 0; i < iters; ++i) {   // 4 independent chains of exp2 -> add -> rcp -> fma (the gate pattern)
        v0 = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v0)), 0.5f, 0.25f);
        v1 = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v1)), 0.5f, 0.25f);
        v2 = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v2)), 0.5f, 0.25f);
        v3 = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v3)), 0.5f, 0.25f);
    }
    return v0 + v1 + v2 + v3;
}
// in-wave interleave: after every MFMA, NF independent plain-VALU fillers, pinned with sched_barrier
template <int KIND, int NF>
__device__ __forceinline__ float do_both(int iters, float seed) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = seed + k;
    float r = 0.f;
    if constexpr (KIND == 0) {
        f4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = seed, y = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
#define FILL _Pragma("unroll") for (int k = 0; k < NF; ++k) { if (k % 3 == 2) v[k] = __builtin_amdgcn_exp2f(v[k]); else v[k] = fmaf(v[k], 0.999f, 0.25f); } __builtin_amdgcn_sched_barrier(0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); FILL
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0); FILL
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); FILL
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0); FILL
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if constexpr (KIND == 3) {     // bf16 32x32x16 with AGPR accumulators (inline asm), 4 accumulators
        bf8 x = {1, 2, 3, 4, 5, 6, 7, (short)seed}, y = {8, 7, 6, 5, 4, 3, 2, 1};
#define CLOB "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
        for (int i = 0; i < iters / 2; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(x), "v"(y) : CLOB); FILL
            asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" :: "v"(x), "v"(y) : CLOB); FILL
            asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" :: "v"(x), "v"(y) : CLOB); FILL
            asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(x), "v"(y) : CLOB); FILL
        }
        r = (float)x[0];
    } else {
        f16v a0 = {}, a1 = {}, a2 = {}, a3 = {};
        bf8 x = {1, 2, 3, 4, 5, 6, 7, (short)seed}, y = {8, 7, 6, 5, 4, 3, 2, 1};
        for (int i = 0; i < iters / 2; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0); FILL
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0); FILL
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0); FILL
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0); FILL
        }
        r = a0[0] + a1[5] + a2[1] + a3[2];
    }
    for (int k = 0; k < 8; ++k) r += v[k];
    return r;
}
template <int KIND, int NF>
__global__ void __launch_bounds__(512) bench_both(int iters, float* out) {
    float r = do_both<KIND, NF>(iters, (float)threadIdx.x);
    if (r == 12345.678f) out[0] = r;
}
template <int KIND, int NF> void run_both(const char* name, int iters) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads = 512; threads >= 256; threads -= 256) {
    bench_both<KIND, NF><<<256, threads>>>(iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) bench_both<KIND, NF><<<256, threads>>>(iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s, %d fillers per MFMA (%d waves/SIMD): %.3f ms\n", name, NF, threads / 256, ms / 5);
    }
}
template <int KIND>
__global__ void __launch_bounds__(512) bench(int mode, int mfma_iters, int valu_iters, float* out) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    // modes 3/4: as mode 2 with s_setprio 3 on the VALU waves / on the MFMA waves; 5: VALU on the OLDER waves
    const bool split = mode >= 2;
    const bool mf = mode == 0 || (split && (mode == 5 ? wave >= 4 : wave < 4));
    const bool va = mode == 1 || (split && (mode == 5 ? wave < 4 : wave >= 4));
    if (mode == 3 && __builtin_amdgcn_readfirstlane(va ? 1 : 0)) __builtin_amdgcn_s_setprio(3);
    if (mode == 4 && __builtin_amdgcn_readfirstlane(mf ? 1 : 0)) __builtin_amdgcn_s_setprio(3);
    if (mf) r += do_mfma<KIND>(mfma_iters, (float)threadIdx.x);
    if (va) r += (valu_iters < 0) ? do_fma(-valu_iters, (float)threadIdx.x * 1e-3f) : do_valu(valu_iters, (float)threadIdx.x * 1e-3f);
    if (r == 12345.678f) out[0] = r;
}
template <int KIND> void run(const char* name, int mfma_iters, int valu_iters) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        bench<KIND><<<256, 512>>>(mode, mfma_iters, valu_iters, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) bench<KIND><<<256, 512>>>(mode, mfma_iters, valu_iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s mode %d: %.3f ms\n", name, mode, ms / 5);
    }
}
int main() {
    // per-wave MFMA work sized to ~equal per-wave VALU work
    run<0>("f32  16x16x4  + exp/rcp", 20000, 26000);
    run<1>("bf16 32x32x16 + exp/rcp", 40000, 26000);
    run<0>("f32  16x16x4  + fma    ", 20000, -40000);
    run<1>("bf16 32x32x16 + fma    ", 40000, -40000);
    run<2>("f32  AGPR acc + exp/rcp", 20000, 26000);
    run<3>("bf16 AGPR acc + exp/rcp", 40000, 26000);
    run<2>("f32  AGPR acc + fma    ", 20000, -40000);
    run<3>("bf16 AGPR acc + fma    ", 40000, -40000);
    run_both<1, 0>("bf16 VGPR acc x4", 40000); run_both<1, 2>("bf16 VGPR acc x4", 40000);
    run_both<1, 4>("bf16 VGPR acc x4", 40000); run_both<1, 6>("bf16 VGPR acc x4", 40000);
    run_both<3, 0>("bf16 AGPR acc x4", 40000); run_both<3, 2>("bf16 AGPR acc x4", 40000);
    run_both<3, 4>("bf16 AGPR acc x4", 40000); run_both<3, 6>("bf16 AGPR acc x4", 40000);
    return 0;
}
