// tn_gemm.h - host side of the two fixed-order reductions of the gradient: the weight-gradient product  dW += P^T Q
// (grad_kernels.h: tn_gemm_kernel + tn_reduce_kernel) and the head rows (head_reduce_kernel).
#pragma once
#include <algorithm>

#include "grad_kernels.h"
#include "handle.h"

namespace rnnwf {

// One product over QT column tiles of Q that start at Q / dW (row strides q_stride / dw_stride elements).
template <typename T, int PT, int QT>
inline int tn_gemm_launch_cols(rnnwf_handle* h, const T* P, const T* Q, int64_t R, T* dW, int q_stride, int dw_stride) {
    int64_t rpb = (R + (int64_t)h->cu_count * 4 - 1) / ((int64_t)h->cu_count * 4);
    rpb = std::max<int64_t>(64, ((rpb + 3) / 4) * 4);
    const unsigned gblocks = (unsigned)((R + rpb - 1) / rpb);
    if (int rc = ensure(h, h->gradPart, (size_t)gblocks * PT * QT * 256 * sizeof(T))) return rc;
    tn_gemm_kernel<T, PT, QT><<<gblocks, TnGemmShape<T, PT, QT>::WAVES * 64, 0, h->stream>>>(P, Q, R, rpb, (T*)h->gradPart.p, q_stride);
    tn_reduce_kernel<T, PT, QT><<<PT * QT, 1024, 0, h->stream>>>((const T*)h->gradPart.p, (int)gblocks, dW, dw_stride);
    RNNWF_HIP(h, hipGetLastError());
    return 0;
}

// P [R][16 PT], Q [R][16 QT] row-major on the device; dW [16 PT][16 QT] receives the sum over all R rows (added to what it holds).
// Timer 4 brackets the launches.
// Wide products (133..260 units; f64 from 53) are taken one 16-byte CHUNK of Q columns at a time (RowChunks: VW tiles, the row's
// last chunk narrower): with all QT tiles at once the busiest wave would hold tiles x QT accumulator fragments - 816 registers at 260
// units - and the kernel spilled them (77 ms per gradient at config 2's size where the arithmetic takes 5; profiles/r04_m_wide_widths.txt).
// The slices partition the columns, the rows of every output element are summed in the same order: same bits as the one-launch form.
template <typename T, int PT, int QT>
inline int tn_gemm_launch(rnnwf_handle* h, const T* P, const T* Q, int64_t R, T* dW) {
    if (R <= 0) return 0;
    using S = TnGemmShape<T, PT, QT>;
    using CQ = RowChunks<T, QT>;
    constexpr int ACC_REGS = tn_max_tiles(PT, S::CW, S::WAVES) * QT * 4 * ((int)sizeof(T) / 4);     // accumulators of the busiest wave
    TimedLaunch tl(h, 4);
    if constexpr (ACC_REGS <= 208 || CQ::NC == 1) {
        return tn_gemm_launch_cols<T, PT, QT>(h, P, Q, R, dW, CQ::COLS, CQ::COLS);
    } else {
        for (int c = 0; c < CQ::NC - 1; ++c)
            if (int rc = tn_gemm_launch_cols<T, PT, CQ::VW>(h, P, Q + 16 * CQ::VW * c, R, dW + 16 * CQ::VW * c, CQ::COLS, CQ::COLS)) return rc;
        return tn_gemm_launch_cols<T, PT, CQ::WL>(h, P, Q + 16 * CQ::VW * (CQ::NC - 1), R, dW + 16 * CQ::VW * (CQ::NC - 1), CQ::COLS, CQ::COLS);
    }
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
