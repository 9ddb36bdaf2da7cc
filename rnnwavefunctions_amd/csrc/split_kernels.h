// split_kernels.h - flip pass of the positive RNN on the bf16x3 engine (split_core.h): same work items and
// outputs as prnn_flip_kernel (gru_kernels.h), 32 chains per wave.  Hidden-state checkpoints are read in the
// layout the f32 base pass wrote them ([N-1][ns/16][KT16][64], unit 4 kt + q of chain c16 at lane (q << 4) | c16).
#pragma once
#include "gru_kernels.h"
#include "split_core.h"

namespace rnnwf {

template <int NF32, int RJ, int WAVES, int MODE>
__global__ void __launch_bounds__(WAVES * 64) prnn_flip_split_kernel(PrnnArgs a, const void* wsplit, int kt16) {
    using C = SplitCore<NF32, RJ, 1, MODE>;
    using L = typename C::L;
    constexpr int NU = C::NU, NR = C::NR;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, wsplit);
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int64_t nsb32 = (a.ns + 31) / 32;
    const int64_t ntiles = (int64_t)(N - 1) * nsb32;
    const float* hck = reinterpret_cast<const float*>(a.hck);
    for (int64_t tile = gw; tile < ntiles; tile += nw) {
        const int i = (int)(tile / nsb32);
        const int64_t sb = tile - (int64_t)i * nsb32;
        const int64_t s = sb * 32 + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        float h[NU];
        {
            const float* src = hck + (((int64_t)i * a.nsb + (sc >> 4)) * kt16) * 64 + (sc & 15);
#pragma unroll
            for (int e = 0; e < NU; ++e) {
                constexpr int dummy = 0; (void)dummy;
                const int u0 = L::unit_of(e, 0), u1 = L::unit_of(e, 1);
                const int u = hh ? u1 : u0;
                h[e] = u < 4 * kt16 ? src[(u >> 2) * 64 + ((u & 3) << 4)] : 0.0f;
            }
        }
        auto spin = [&](int n) { return (int)((a.bits[(int64_t)(n >> 5) * a.ns + sc] >> (n & 31)) & 1); };
        int sig_in = 1 - spin(i);
        double lp = 0.0;
        unsigned R[3][NR];
        for (int n = i + 1; n < N; ++n) {
            const int sig = spin(n);
            C::split(h, sig_in, R);
            C::step(lds, sig_in, R, h, lane);
            float z[1];
            C::head(lds, h, lane, z);
            float lp0, lp1;
            log_softmax2(z[0], lp0, lp1);
            lp += (double)(sig ? lp1 : lp0);
            sig_in = sig;
        }
        if (valid && hh == 0) {
            const int64_t row = a.row_of_pos ? a.row_of_pos[i] : i + 1;
            a.lpq[row * a.ns + s] += lp;
        }
    }
}

}  // namespace rnnwf
