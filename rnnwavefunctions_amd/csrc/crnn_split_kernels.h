// crnn_split_kernels.h - J1-J2 swap pass of the complex RNN on the bf16x3 engine (split_core.h).
#pragma once
#include "crnn_kernels.h"
#include "split_core.h"

namespace rnnwf {

// J1-J2 swap pass on the bf16x3 engine: same items / outputs as crnn_swap_kernel, 32 items per wave tile
// (the tile table must have been scanned with tile_items = 32).
template <int NF32, int RJ, int WAVES, int MODE>
__global__ void __launch_bounds__(WAVES * 64, (3 * NF32 + (3 * RJ + 15) / 16) <= 5 ? 2 : 1) crnn_swap_split_kernel(CrnnArgs a, const void* wsplit, int kt16) {
    using C = SplitCore<NF32, RJ, 3, MODE>;
    using L = typename C::L;
    constexpr int NU = C::NU, NR = C::NR;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    C::stage(lds, wsplit);
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int64_t ntiles = a.tile_start[N];
    const float* hck = reinterpret_cast<const float*>(a.hck);
    for (int64_t tile = gw; tile < ntiles; tile += nw) {
        int lo = 0;
        {
            int l = 0, r = N;
            while (r - l > 1) {
                const int mid = (l + r) >> 1;
                if (a.tile_start[mid] <= tile) l = mid; else r = mid;
            }
            lo = l;
        }
        const int k = (int)(tile - a.tile_start[lo]) * 32 + c;
        const bool valid = k < a.cnt[lo];
        const SwapItem it = a.items[(int64_t)lo * a.cap + (valid ? k : 0)];
        const int64_t s = it.s;
        float h[NU];
        {
            const float* src = hck + (((int64_t)lo * a.nsb + (s >> 4)) * kt16) * 64 + (s & 15);
#pragma unroll
            for (int e = 0; e < NU; ++e) {
                const int u = hh ? L::unit_of(e, 1) : L::unit_of(e, 0);
                h[e] = u < 4 * kt16 ? src[(u >> 2) * 64 + ((u & 3) << 4)] : 0.0f;
            }
        }
        int num_up = 0;
        for (int w = 0; w < (lo >> 5); ++w) num_up += __popc(a.bits[(int64_t)w * a.ns + s]);
        uint32_t word = a.bits[(int64_t)(lo >> 5) * a.ns + s];
        num_up += __popc(word & ((1u << (lo & 31)) - 1u));
        int sig_in = 1 - (int)((word >> (lo & 31)) & 1);
        num_up += sig_in;
        double re = 0.0, im = 0.0;
        unsigned R[3][NR];
        for (int n = lo + 1; n < N; ++n) {
            if ((n & 31) == 0) word = a.bits[(int64_t)(n >> 5) * a.ns + s];
            C::split(h, sig_in, R);
            C::step(lds, sig_in, R, h, lane);
            float z[3];
            C::head(lds, h, lane, z);
            float la0, la1, w0, ph0, ph1;
            crnn_site(z, n, N, num_up, la0, la1, w0, ph0, ph1);
            const int sig = (int)((word >> (n & 31)) & 1) ^ (n == it.hi ? 1 : 0);
            re += (double)(sig ? la1 : la0);
            im += (double)(sig ? ph1 : ph0);
            num_up += sig;
            sig_in = sig;
        }
        if (valid && hh == 0) {
            const double2 b = a.cb[(int64_t)lo * a.ns + s];
            const double2 t = a.tot[s];
            const double dre = b.x + re - t.x, dim = b.y + im - t.y;
            const double mag = exp(dre) * (double)it.coef;
            a.contrib[s * (2 * N) + it.slot] = make_double2(mag * cos(dim), mag * sin(dim));
        }
    }
}

}  // namespace rnnwf
