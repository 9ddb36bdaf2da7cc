// grad_wide.hip - the backward kernels of the wide shapes (one wave per SIMD, 512 registers): positive and complex RNN from 101 units,
// float64 GRU 69..100 units.
// In grad.hip's translation unit (compiled with -amdgpu-mfma-vgpr-form, build.py) hipcc 7.2 crashes on several of these in
// 'AMDGPU Rewrite AGPR-Copy-MFMA' - which ones changes with unrelated edits of the kernel; rounds 2-3 ran three of them with 8 waves per
// workgroup instead - 256 registers per wave, 1.3 - 2.7 KB of spills per lane (the float64 GRU's backward kernel at 100 units: 12 ms
// where the 68-unit one takes 1.6).  This unit is compiled WITHOUT that option (accumulators in AGPRs), which the pass survives.
#include "grad_kernels.h"
#include "models.h"

namespace rnnwf {

const void* grad_wide_kernel(int which) {
    switch (which) {
        case 0: return (const void*)gru_bwd_kernel<float, 8, 4, 3>;
        case 1: return (const void*)gru_bwd_kernel<float, 12, 4, 3>;
        case 2: return (const void*)gru_bwd_kernel<double, 6, 4, 1>;
        case 3: return (const void*)gru_bwd_kernel<float, 16, 4, 3>;
        case 4: return (const void*)gru_bwd_kernel<float, 8, 4, 1>;
        case 5: return (const void*)gru_bwd_kernel<float, 12, 4, 1>;
        case 6: return (const void*)gru_bwd_kernel<float, 16, 4, 1>;
    }
    return nullptr;
}

void grad_wide_launch(int which, unsigned grid, size_t lds, hipStream_t stream, const GradArgs& a) {
    switch (which) {
        case 0: gru_bwd_kernel<float, 8, 4, 3><<<grid, 256, lds, stream>>>(a); break;
        case 1: gru_bwd_kernel<float, 12, 4, 3><<<grid, 256, lds, stream>>>(a); break;
        case 2: gru_bwd_kernel<double, 6, 4, 1><<<grid, 256, lds, stream>>>(a); break;
        case 3: gru_bwd_kernel<float, 16, 4, 3><<<grid, 256, lds, stream>>>(a); break;
        case 4: gru_bwd_kernel<float, 8, 4, 1><<<grid, 256, lds, stream>>>(a); break;
        case 5: gru_bwd_kernel<float, 12, 4, 1><<<grid, 256, lds, stream>>>(a); break;
        case 6: gru_bwd_kernel<float, 16, 4, 1><<<grid, 256, lds, stream>>>(a); break;
    }
}

}  // namespace rnnwf
