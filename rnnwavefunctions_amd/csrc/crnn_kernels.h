// crnn_kernels.h - complex RNN wave function with the U(1) zero-magnetisation mask and the J1-J2
// local-energy path (J1J2/ComplexRNNwavefunction.py, J1J2/TrainingRNN_J1J2.py).
//
//   crnn_base_kernel    : masked ancestral sampling (:45-103) or teacher-forced log-amplitude (:105-169),
//                         16 chains per wave; optional checkpoints of the hidden state after every site and
//                         "swap base" values  cb[n][s] = sum_{m<n} log psi_m(s_m) + log psi_n(1 - s_n).
//   j1j2_enumerate_kernel: connected configurations of J1J2MatrixElements (TrainingRNN_J1J2.py:12-93) on the
//                         device: one item per anti-aligned bond, compacted per first-changed site `lo` with a
//                         wavefront ballot + one atomic per wave; also the diagonal matrix element.
//   j1j2_tile_scan_kernel: prefix sum of the per-`lo` tile counts (tiles of 16 items, longest chains first).
//   crnn_swap_kernel    : for every item re-evaluates sites lo+1..N-1 of the swapped configuration starting
//                         from the checkpoint of site lo, and writes  H_k exp(log psi(s') - log psi(s)).
//   j1j2_eloc_kernel    : E_loc[s] = diag + sum_k contributions, in the reference's bond order (:277-279).
#pragma once
#include "gru_kernels.h"

namespace rnnwf {

struct SwapItem {      // 16 bytes
    int32_t s;         // sample index within the launch
    int32_t hi;        // second changed site (> lo)
    int32_t slot;      // bond slot in the per-sample contribution row (J1 bond a -> a, J2 bond a -> N + a)
    float coef;        // matrix element  +-J/2
};

struct CrnnArgs {
    const void* wimg;
    const void* wbf;           // bf16x3 A fragments of the cooperative base pass (BaseBfLayout), nullptr: none
    int32_t N;
    int64_t ns, nsb;
    uint32_t* bits;
    void* hck;                 // [N-1][nsb][KT][64] float, nullptr: no checkpoints
    double2* cb;               // [N][ns] swap base, nullptr: none
    double2* tot;              // [ns] log psi(s) (re, im) in f64
    float2* out_amp;           // [ns] complex64 log-amplitude (may be nullptr)
    double* out_logp;          // [ns] 2 Re log psi (may be nullptr)
    uint64_t seed, step;
    int64_t sample_offset;
    int32_t sampling;
    // swap pass
    const int32_t* tile_start; // [N+1]
    const int32_t* cnt;        // [N] items per lo
    const SwapItem* items;     // [N][cap]
    int64_t cap;
    double2* contrib;          // [2N][ns]  (bond slot major: the assembly kernel reads it coalesced)
    const int64_t* rec_start;  // [N] first wave-step record of site lo's tiles (stacked layers on the bf16x3 engine), nullptr otherwise
};

// One site of the complex RNN (ComplexRNNwavefunction.py:83-93,143-157) from the head outputs
// z = (amplitude logit difference, phase logit 0, phase logit 1):
//   amplitudes a = sqrt(softmax)            -> log a_s = 1/2 log p_s
//   for n >= N/2 the U(1) mask zeroes a spin value that would overshoot N/2 and l2-normalises: with both
//   values allowed nothing changes (a0^2 + a1^2 = 1), with one allowed its amplitude becomes 1 (log 0) and the
//   other 0 (log -inf)
//   phase = pi * softsign(phase logit of the chosen spin)
// la0/la1: log-amplitudes, w0: probability of spin 0 for the sampler, ph0/ph1: phases.
__device__ __forceinline__ void crnn_site(const float (&z)[3], int n, int N, int num_up, float& la0, float& la1,
                                          float& w0, float& ph0, float& ph1) {
#pragma clang fp contract(off)
    float lp0, lp1;
    log_softmax2(z[0], lp0, lp1);
    la0 = 0.5f * lp0;
    la1 = 0.5f * lp1;
    w0 = prob0(z[0]);
    if (2 * n >= N) {                                    // n >= N/2: enforce zero magnetisation
        const int base = N / 2 - 1;
        const bool ok_down = base - (n - num_up) >= 0;   // [activations_down, activations_up]
        const bool ok_up = base - num_up >= 0;
        const float ninf = -__builtin_inff();
        if (!ok_down) { la0 = ninf; w0 = 0.0f; if (ok_up) la1 = 0.0f; }
        if (!ok_up) { la1 = ninf; if (ok_down) { la0 = 0.0f; w0 = 1.0f; } }
    }
    ph0 = 3.14159265358979323846f * (z[1] / (1.0f + fabsf(z[1])));   // pi * softsign
    ph1 = 3.14159265358979323846f * (z[2] / (1.0f + fabsf(z[2])));
}

template <int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) crnn_base_kernel(CrnnArgs a) {
    using C = GruCore<float, NFULL, 3>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const char* img = C::stage(lds, a.wimg);       // LDS, or the global image where it exceeds LDS (GruLayout::SPILL)
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    for (int64_t sb = gw; sb < a.nsb; sb += nw) {
        const int64_t s = sb * kChains + c;
        const bool valid = s < a.ns;
        const int64_t sc = valid ? s : a.ns - 1;
        float h[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) h[kt] = 0.0f;
        int sig_in = -1, num_up = 0;
        uint32_t word = 0;
        double re = 0.0, im = 0.0;
        for (int n = 0; n < N; ++n) {
            if (!a.sampling && (n & 31) == 0) word = a.bits[(int64_t)(n >> 5) * a.ns + sc];
            C::template step<true>(img, sig_in, h, lane);    // bias last: bit-identical to crnn_base_coop_kernel
            float z[3];
            C::head(img, h, lane, z);
            float la0, la1, w0, ph0, ph1;
            crnn_site(z, n, N, num_up, la0, la1, w0, ph0, ph1);
            int sig;
            if (a.sampling) {
                // tf.random.categorical(log(a^2)): class 0 iff u * total < a0^2; a masked class has logit -inf
                // and is skipped by TF's kernel - same outcome here since its weight is exactly 0
                const float u = philox_uniform(a.seed, a.step, (uint64_t)(a.sample_offset + sc), n);
                sig = (u < w0) ? 0 : 1;
                word |= (uint32_t)sig << (n & 31);
                if (((n & 31) == 31 || n == N - 1) && valid && q == 0) a.bits[(int64_t)(n >> 5) * a.ns + s] = word;
                if ((n & 31) == 31) word = 0;
            } else {
                sig = (word >> (n & 31)) & 1;
            }
            if (a.cb && valid && q == 0)
                a.cb[(int64_t)n * a.ns + s] = make_double2(re + (double)(sig ? la0 : la1), im + (double)(sig ? ph0 : ph1));
            re += (double)(sig ? la1 : la0);
            im += (double)(sig ? ph1 : ph0);
            if (a.hck && n < N - 1) {
                float* dst = reinterpret_cast<float*>(a.hck) + (((int64_t)n * a.nsb + sb) * KT) * 64 + lane;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) dst[kt * 64] = h[kt];
            }
            num_up += sig;
            sig_in = sig;
        }
        if (valid && q == 0) {
            if (a.tot) a.tot[s] = make_double2(re, im);
            if (a.out_amp) a.out_amp[s] = make_float2((float)re, (float)im);
            if (a.out_logp) a.out_logp[s] = 2.0 * re;
        }
    }
}

// The same pass with NFULL + 1 waves per block of 16 chains (gru_kernels.h, coop_base_pass): used when there are fewer
// blocks than SIMDs.  Bit-identical to crnn_base_kernel.
template <int NFULL, bool BF = false, int NL = 1>
__global__ void __launch_bounds__((NL > 1 ? MlCoopLayout<NFULL, NL, 3>::THREADS : BF ? (NFULL + 2) * 64 * BaseBfLayout<NFULL>::NB : (NFULL + 1) * 64))
crnn_base_coop_kernel(CrnnArgs a) {
    using C = GruCore<float, NFULL, 3>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int N = a.N;
    int64_t s = 0, sc = 0;
    bool valid = false;
    uint32_t word = 0, word_in = 0;
    int num_up = 0;
    double re = 0.0, im = 0.0;
    float u_next = 0.0f;
    auto prefetch = [&](int n) {      // site n's uniform / word of given spins, fetched behind the publication of site n - 1's spin
        if (a.sampling) u_next = philox_uniform(a.seed, a.step, (uint64_t)(a.sample_offset + sc), n);
        else if ((n & 31) == 0) word_in = a.bits[(int64_t)(n >> 5) * a.ns + sc];
    };
    auto begin = [&](int64_t sb) {
            s = sb * kChains + c;
            valid = s < a.ns;
            sc = valid ? s : a.ns - 1;
            word = 0;
            num_up = 0;
            re = im = 0.0;
            prefetch(0);
        };
    auto site = [&](int n, const float (&h)[KT], auto&& publish) {
            float z[3];
            C::head(lds, h, lane, z);
            float la0, la1, w0, ph0, ph1;
            crnn_site(z, n, N, num_up, la0, la1, w0, ph0, ph1);
            int sig;
            if (a.sampling) sig = (u_next < w0) ? 0 : 1;
            else sig = (word_in >> (n & 31)) & 1;
            publish(sig);                                               // the other waves wait for this
            if (a.sampling) {
                word |= (uint32_t)sig << (n & 31);
                if (((n & 31) == 31 || n == N - 1) && valid && q == 0) a.bits[(int64_t)(n >> 5) * a.ns + s] = word;
                if ((n & 31) == 31) word = 0;
            }
            if (a.cb && valid && q == 0)
                a.cb[(int64_t)n * a.ns + s] = make_double2(re + (double)(sig ? la0 : la1), im + (double)(sig ? ph0 : ph1));
            re += (double)(sig ? la1 : la0);
            im += (double)(sig ? ph1 : ph0);
            num_up += sig;
            if (n + 1 < N) prefetch(n + 1);
        };
    auto end = [&](int64_t) {
            if (valid && q == 0) {
                if (a.tot) a.tot[s] = make_double2(re, im);
                if (a.out_amp) a.out_amp[s] = make_float2((float)re, (float)im);
                if (a.out_logp) a.out_logp[s] = 2.0 * re;
            }
        };
    if constexpr (NL > 1) coop_ml_base_pass<NFULL, NL, 3>(lds, a.wimg, N, a.nsb, a.hck, begin, site, end);      // stacked layers (ml_coop.h)
    else if constexpr (BF) coop_base_pass_bf<NFULL, 3>(lds, a.wimg, a.wbf, N, a.nsb, a.hck, begin, site, end);
    else coop_base_pass<NFULL, 3>(lds, a.wimg, N, a.nsb, a.hck, 0, begin, site, end);
}

// ---- connected configurations -----------------------------------------------------------------------
struct J1J2Args {
    const uint32_t* bits;
    int64_t ns;
    int32_t N;
    const double *J1, *J2, *Bz;   // device, (N) each
    int32_t periodic, marshall;
    int32_t* cnt;                 // [N] zeroed before launch
    SwapItem* items;              // [N][cap]
    int64_t cap;
    double2* contrib;             // [2N][ns] (bond slot major), every entry zeroed here; the swap pass overwrites the active bonds
    double* diag;                 // [ns]
};

__device__ __forceinline__ int spin_of(const uint32_t* bits, int64_t ns, int64_t s, int p) {
    return (int)((bits[(int64_t)(p >> 5) * ns + s] >> (p & 31)) & 1);
}

// first changed site of bond slot `slot` (J1 bond a -> slot a, J2 bond a -> slot N + a)
__device__ __forceinline__ int lo_of_block(int N, int slot, int periodic) {
    const int dist = slot < N ? 1 : 2, site = slot < N ? slot : slot - N;
    const int t = (site + dist) % N;
    (void)periodic;
    return site < t ? site : t;
}

// grid.y = bond slot (0..2N-1), 256 samples per block
static __global__ void __launch_bounds__(256) j1j2_enumerate_kernel(J1J2Args a) {
    const int N = a.N;
    const int slot = blockIdx.y;
    const int dist = slot < N ? 1 : 2;
    const int site = slot < N ? slot : slot - N;
    const int lim = a.periodic ? N : N - dist;
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = s < a.ns;
    const double J = site < lim ? (dist == 1 ? a.J1[site] : a.J2[site]) : 0.0;
    bool active = false;
    int lo = 0, hi = 0;
    if (in_range && site < lim && J != 0.0) {
        const int t = (site + dist) % N;
        lo = site < t ? site : t;
        hi = site < t ? t : site;
        active = spin_of(a.bits, a.ns, s, lo) != spin_of(a.bits, a.ns, s, hi);
    }
    // contrib [2N][ns] starts as zeros (inactive bonds contribute 0; the swap pass overwrites the active ones): with the bond-slot-major
    // layout this thread's element is its wave's next 16 bytes - one coalesced store here instead of a 12.8 MB memset launch per step
    // (round 1's sample-major layout made these stores cost 80 of this kernel's 107 us; hence the memset that stood here until round 4)
    if (in_range) a.contrib[(int64_t)slot * a.ns + s] = make_double2(0.0, 0.0);
    // lo depends on the slot only, i.e. it is uniform over the block: ballot-ranked slots inside each wave, the four
    // waves' counts combined in LDS, ONE atomic per block (same-address atomics serialise at L2: 314 per counter
    // with one per wave at config 3, 79 now)
    __shared__ int wave_cnt[4];
    __shared__ int block_base;
    const unsigned long long mask = __ballot(active);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        block_base = total ? atomicAdd(&a.cnt[lo_of_block(N, slot, a.periodic)], total) : 0;
    }
    __syncthreads();
    {
        int base = block_base;
        for (int w = 0; w < wave; ++w) base += wave_cnt[w];
        if (active) {
            const int k = base + __popcll(mask & ((1ull << lane) - 1ull));
            SwapItem it;
            it.s = (int32_t)s;
            it.hi = hi;
            it.slot = slot;
            it.coef = (float)((dist == 1 && a.marshall) ? -J / 2 : J / 2);
            a.items[(int64_t)lo * a.cap + k] = it;
        }
    }
    __shared__ double cpl[3][256];   // slot-0 blocks: J1, J2, Bz
    if (slot == 0 && N <= 256) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) { cpl[0][i] = a.J1[i]; cpl[1][i] = a.J2[i]; cpl[2][i] = a.Bz[i]; }
        __syncthreads();
    }
    if (slot == 0 && in_range && N <= 256) {     // diagonal element, once per sample (TrainingRNN_J1J2.py:32,46-57)
        // These 40 blocks were the kernel's long pole: 3 N iterations, each waiting for a scalar load of its coupling and two
        // global loads of spins.  Now: the sample's packed words in registers (N <= 256) and the couplings in LDS (staged by the
        // whole block below); the same three sums in the same order.
        constexpr int MAXW = 8;
        uint32_t wd[MAXW];
        const int W = (N + 31) / 32;
#pragma unroll
        for (int w = 0; w < MAXW; ++w) wd[w] = w < W ? a.bits[(int64_t)w * a.ns + s] : 0u;
        auto sp = [&](int i) -> int {
            uint32_t v = 0;
#pragma unroll
            for (int w = 0; w < MAXW; ++w) if (w == (i >> 5)) v = wd[w];
            return (int)((v >> (i & 31)) & 1);
        };
        double d = 0.0;
        for (int i = 0; i < N; ++i) d += ((double)sp(i) - 0.5) * cpl[2][i];
        const int lim1 = a.periodic ? N : N - 1, lim2 = a.periodic ? N : N - 2;
        for (int i = 0; i < lim1; ++i) d += (sp(i) != sp((i + 1) % N) ? -0.25 : 0.25) * cpl[0][i];
        for (int i = 0; i < lim2; ++i)
            if (cpl[1][i] != 0.0) d += (sp(i) != sp((i + 2) % N) ? -0.25 : 0.25) * cpl[1][i];
        a.diag[s] = d;
    } else if (slot == 0 && in_range) {      // chains longer than 256 sites: straight from global memory
        double d = 0.0;
        for (int i = 0; i < N; ++i) d += ((double)spin_of(a.bits, a.ns, s, i) - 0.5) * a.Bz[i];
        const int lim1 = a.periodic ? N : N - 1, lim2 = a.periodic ? N : N - 2;
        for (int i = 0; i < lim1; ++i)
            d += (spin_of(a.bits, a.ns, s, i) != spin_of(a.bits, a.ns, s, (i + 1) % N) ? -0.25 : 0.25) * a.J1[i];
        for (int i = 0; i < lim2; ++i)
            if (a.J2[i] != 0.0)
                d += (spin_of(a.bits, a.ns, s, i) != spin_of(a.bits, a.ns, s, (i + 2) % N) ? -0.25 : 0.25) * a.J2[i];
        a.diag[s] = d;
    }
}

// tile_start[lo] = sum_{l < lo} ceil(cnt[l] / 16); tile_start[N] = total;
// totals[0] = sum cnt (off-diagonal configurations), totals[1] = sum cnt[lo] (N-1-lo) (cell evaluations),
// totals[2] = sum tiles[lo] (N-1-lo) (wave-steps actually issued)
// One wave: lane l takes the sites l, l + 64, ... (one load each instead of N dependent ones), wave-level prefix sums.
// totals_host: the same three numbers straight into pinned host memory (read after the caller's stream sync; no copy launch).
// rec_start (optional; stacked layers on the bf16x3 engine): rec_start[lo] = sum_{l < lo} tiles[l] (N-1-l), the first wave-step record of
// the tiles of site lo in the layer pipeline's record buffers (split_kernels.h).
static __global__ void __launch_bounds__(64) j1j2_tile_scan_kernel(const int32_t* cnt, int N, int32_t* tile_start, int64_t* totals,
                                                                   int64_t* totals_host, int tile_items, int64_t* rec_start = nullptr) {
    const int lane = threadIdx.x;
    int32_t carry = 0;
    int64_t rcarry = 0;
    int64_t items = 0, evals = 0, wsteps = 0;
    for (int base = 0; base < N; base += 64) {
        const int lo = base + lane;
        const int c = lo < N ? cnt[lo] : 0;
        const int t = (c + tile_items - 1) / tile_items;
        int incl = t;                                        // inclusive prefix sum of the tile counts over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lo < N) tile_start[lo] = carry + incl - t;
        carry += __shfl(incl, 63);
        if (rec_start) {                                     // the same scan over tiles x chain length
            const int64_t w = lo < N ? (int64_t)t * (N - 1 - lo) : 0;
            int64_t rincl = w;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int64_t v = __shfl_up(rincl, d);
                if (lane >= d) rincl += v;
            }
            if (lo < N) rec_start[lo] = rcarry + rincl - w;
            rcarry += __shfl(rincl, 63);
        }
        if (lo < N) {
            items += c;
            evals += (int64_t)c * (N - 1 - lo);
            wsteps += (int64_t)t * (N - 1 - lo);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        items += __shfl_xor(items, d);
        evals += __shfl_xor(evals, d);
        wsteps += __shfl_xor(wsteps, d);
    }
    if (lane == 0) {
        tile_start[N] = carry;
        totals[0] = items; totals[1] = evals; totals[2] = wsteps;
        if (totals_host) { totals_host[0] = items; totals_host[1] = evals; totals_host[2] = wsteps; }
    }

}

template <int NFULL, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) crnn_swap_kernel(CrnnArgs a) {
    using C = GruCore<float, NFULL, 3>;
    constexpr int KT = C::KT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const char* img = C::stage(lds, a.wimg);       // LDS, or the global image where it exceeds LDS (GruLayout::SPILL)
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * WAVES;
    const int N = a.N;
    const int64_t ntiles = a.tile_start[N];
    for (int64_t tile = gw; tile < ntiles; tile += nw) {
        int lo = 0;                                   // largest lo with tile_start[lo] <= tile (wave-uniform)
        {
            int l = 0, r = N;
            while (r - l > 1) {
                const int mid = (l + r) >> 1;
                if (a.tile_start[mid] <= tile) l = mid; else r = mid;
            }
            lo = l;
        }
        const int k = (int)(tile - a.tile_start[lo]) * kChains + c;
        const bool valid = k < a.cnt[lo];
        const SwapItem it = a.items[(int64_t)lo * a.cap + (valid ? k : 0)];
        const int64_t s = it.s;
        float h[KT];
        {
            const float* src = reinterpret_cast<const float*>(a.hck) +
                               (((int64_t)lo * a.nsb + (s >> 4)) * KT) * 64 + (q << 4) + (s & 15);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) h[kt] = src[kt * 64];
        }
        int num_up = 0;                               // ups among sites < lo, then the swapped spin at lo
        for (int w = 0; w < (lo >> 5); ++w) num_up += __popc(a.bits[(int64_t)w * a.ns + s]);
        uint32_t word = a.bits[(int64_t)(lo >> 5) * a.ns + s];
        num_up += __popc(word & ((1u << (lo & 31)) - 1u));
        int sig_in = 1 - (int)((word >> (lo & 31)) & 1);
        num_up += sig_in;
        double re = 0.0, im = 0.0;
        for (int n = lo + 1; n < N; ++n) {
            if ((n & 31) == 0) word = a.bits[(int64_t)(n >> 5) * a.ns + s];
            C::step(img, sig_in, h, lane);
            float z[3];
            C::head(img, h, lane, z);
            float la0, la1, w0, ph0, ph1;
            crnn_site(z, n, N, num_up, la0, la1, w0, ph0, ph1);
            const int sig = (int)((word >> (n & 31)) & 1) ^ (n == it.hi ? 1 : 0);
            re += (double)(sig ? la1 : la0);
            im += (double)(sig ? ph1 : ph0);
            num_up += sig;
            sig_in = sig;
        }
        if (valid && q == 0) {
            const double2 b = a.cb[(int64_t)lo * a.ns + s];
            const double2 t = a.tot[s];
            const double dre = b.x + re - t.x, dim = b.y + im - t.y;
            const double mag = exp(dre) * (double)it.coef;
            a.contrib[(int64_t)it.slot * a.ns + s] = make_double2(mag * cos(dim), mag * sin(dim));
        }
    }
}

// E_loc[s] = diag + sum over bond slots (J1 bonds by site, then J2 bonds by site: the reference's row order)
// contrib is [2N][ns]: consecutive samples read consecutive 16-byte values; the sum runs in the reference's bond order.
// cnt_to_clear: the per-site item counters of this step, zeroed here for the next one (saves a memset launch per step).
static __global__ void j1j2_eloc_kernel(const double2* __restrict__ contrib, const double* __restrict__ diag, int64_t ns,
                                 int N, float2* __restrict__ eloc, int32_t* cnt_to_clear) {
    if (cnt_to_clear && blockIdx.x == 0 && (int)threadIdx.x < N) cnt_to_clear[threadIdx.x] = 0;
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    double re = diag[s], im = 0.0;
    int k = 0;
    for (; k + 8 <= 2 * N; k += 8) {                     // eight loads in flight, added in bond order
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = contrib[(int64_t)(k + u) * ns + s];
#pragma unroll
        for (int u = 0; u < 8; ++u) { re += v[u].x; im += v[u].y; }
    }
    for (; k < 2 * N; ++k) {
        const double2 v = contrib[(int64_t)k * ns + s];
        re += v.x;
        im += v.y;
    }
    eloc[s] = make_float2((float)re, (float)im);
}

}  // namespace rnnwf
