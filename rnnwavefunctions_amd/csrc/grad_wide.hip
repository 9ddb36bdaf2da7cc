// grad_wide.hip - the backward kernels of three wide shapes in their 4-wave form (one wave per SIMD, 512 registers):
//     complex RNN 101..132 and 133..196 units, float64 GRU 69..100 units.
// In grad.hip's translation unit (compiled with -amdgpu-mfma-vgpr-form, build.py) hipcc 7.2 crashes on exactly these three in
// 'AMDGPU Rewrite AGPR-Copy-MFMA'; rounds 2-3 therefore ran them with 8 waves per workgroup - 256 registers per wave, 1.3 - 2.7 KB
// of spills per lane (the float64 GRU's backward kernel at 100 units: 12 ms where the 68-unit one takes 1.6).  This unit is compiled
// WITHOUT that option (accumulators in AGPRs), which the pass survives.
#include "grad_kernels.h"
#include "models.h"

namespace rnnwf {

const void* grad_wide_kernel(int which) {
    switch (which) {
        case 0: return (const void*)gru_bwd_kernel<float, 8, 4, 3>;
        case 1: return (const void*)gru_bwd_kernel<float, 12, 4, 3>;
        case 2: return (const void*)gru_bwd_kernel<double, 6, 4, 1>;
    }
    return nullptr;
}

void grad_wide_launch(int which, unsigned grid, size_t lds, hipStream_t stream, const GradArgs& a) {
    switch (which) {
        case 0: gru_bwd_kernel<float, 8, 4, 3><<<grid, 256, lds, stream>>>(a); break;
        case 1: gru_bwd_kernel<float, 12, 4, 3><<<grid, 256, lds, stream>>>(a); break;
        case 2: gru_bwd_kernel<double, 6, 4, 1><<<grid, 256, lds, stream>>>(a); break;
    }
}

}  // namespace rnnwf
