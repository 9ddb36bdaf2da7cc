// tn_gemm.h - host side of the two fixed-order reductions of the gradient: the weight-gradient product  dW += P^T Q
// (grad_kernels.h: tn_gemm_kernel + tn_reduce_kernel) and the head rows (head_reduce_kernel).
#pragma once
#include <algorithm>

#include "grad_kernels.h"
#include "handle.h"

namespace rnnwf {

// P [R][16 PT], Q [R][16 QT] row-major on the device; dW [16 PT][16 QT] receives the sum over all R rows (added to what it holds).
// Timer 4 brackets both launches.
template <typename T, int PT, int QT>
inline int tn_gemm_launch(rnnwf_handle* h, const T* P, const T* Q, int64_t R, T* dW) {
    if (R <= 0) return 0;
    int64_t rpb = (R + (int64_t)h->cu_count * 4 - 1) / ((int64_t)h->cu_count * 4);
    rpb = std::max<int64_t>(64, ((rpb + 3) / 4) * 4);
    const unsigned gblocks = (unsigned)((R + rpb - 1) / rpb);
    if (int rc = ensure(h, h->gradPart, (size_t)gblocks * PT * QT * 256 * sizeof(T))) return rc;
    {
        TimedLaunch tl(h, 4);
        tn_gemm_kernel<T, PT, QT><<<gblocks, TnGemmShape<T, PT, QT>::WAVES * 64, 0, h->stream>>>(P, Q, R, rpb, (T*)h->gradPart.p);
        tn_reduce_kernel<T, PT, QT><<<PT * QT, 1024, 0, h->stream>>>((const T*)h->gradPart.p, (int)gblocks, dW);
    }
    RNNWF_HIP(h, hipGetLastError());
    return 0;
}

// Room for one head row of n entries per wave of a backward pass of `waves` waves; call before the launch, pass the pointer on.
template <typename T>
inline int head_part_alloc(rnnwf_handle* h, size_t waves, int n, T** out) {
    if (int rc = ensure(h, h->gradHeadPart, waves * (size_t)n * sizeof(T))) return rc;
    *out = (T*)h->gradHeadPart.p;
    return 0;
}
// head_grad[j] += sum over the waves' rows (grad_kernels.h: head_reduce_kernel), right behind the backward pass on the stream.
template <typename T>
inline void head_reduce_launch(rnnwf_handle* h, size_t waves, int n, T* head_grad) {
    head_reduce_kernel<T><<<(n + 63) / 64, 1024, 0, h->stream>>>((const T*)h->gradHeadPart.p, (int)waves, n, head_grad);
}

}  // namespace rnnwf
