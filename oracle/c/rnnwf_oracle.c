/*
 * rnnwf_oracle.c - plain-C restatement of the reference's 1D pRNN / TFIM hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ as a second checker and by
 * bench.py's cpu_baseline leg ("port").  Never linked into or loaded by the product.
 *
 * It executes the REFERENCE FORMULATION literally:
 *   - RNNwavefunction.log_probability  1DTFIM/RNNwavefunction.py:76-118  (f32 cell, f64 log-sum)
 *   - RNNwavefunction.sample           1DTFIM/RNNwavefunction.py:35-74   (explicit uniforms)
 *   - Ising_local_energies             1DTFIM/TrainingRNN_1DTFIM.py:13-75: builds the (N+1, ns, N)
 *     queue of flipped configurations and scores every one of them from site 0 in chunks of at most
 *     25000 rows; no hidden-state prefix reuse.
 * The cuDNN-compatible GRU cell follows TF 1.13.1.  Rows are processed in
 * register blocks of RB chains so that every weight row is reused RB times (what a batched SGEMM
 * does); OpenMP spreads the blocks over the host cores.  exp/tanh are branch-free polynomial
 * versions (Cephes expf, ~1 ulp) so that gcc vectorises the gate loops, as Eigen does inside TF.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RB 16

typedef struct {
    int H;
    const float *Wg, *bg, *Wci, *bci, *Wch, *bch, *Wd, *bd;
} gru_params;

static inline float expf_poly(float x) {
    /* Cephes expf: x = n ln2 + r, exp(r) by a degree-5 polynomial; clamped, branch-free */
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    float fx = floorf(x * 1.44269504088896341f + 0.5f);
    float r = x - fx * 0.693359375f;
    r = r - fx * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * r * r + r + 1.0f;
    union { int32_t i; float f; } u;
    u.i = ((int32_t)fx + 127) << 23;
    return p * u.f;
}
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf_poly(-x)); }
static inline float tanhf_(float x) {
    float e = expf_poly(-2.0f * fabsf(x));
    float t = (1.0f - e) / (1.0f + e);
    return copysignf(t, x);
}

/* One GRU step + Dense(2) + softmax for a block of nb <= RB chains.
 * h: [RB][H] (in/out).  sig[i]: input spin of chain i (-1: zero vector).  p: [RB][2] (out). */
static void gru_block_step(const gru_params* P, int nb, float* h, const int* sig, float* p, float* scratch) {
    const int H = P->H;
    float* g = scratch;               /* [RB][2H] gate pre-activations */
    float* ch = scratch + RB * 2 * H; /* [RB][H]  candidate hidden projection */
    for (int i = 0; i < nb; ++i) {
        float* gi = g + i * 2 * H;
        float* ci = ch + i * H;
        if (sig[i] >= 0) {
            const float* wrow = P->Wg + (size_t)sig[i] * 2 * H;
            for (int j = 0; j < 2 * H; ++j) gi[j] = wrow[j];
        } else {
            for (int j = 0; j < 2 * H; ++j) gi[j] = 0.0f;
        }
        for (int j = 0; j < H; ++j) ci[j] = 0.0f;
    }
    for (int k = 0; k < H; ++k) {     /* [x,h] Wg and h Wch: weight row k reused by all nb chains */
        const float* wg = P->Wg + (size_t)(2 + k) * 2 * H;
        const float* wc = P->Wch + (size_t)k * H;
        for (int i = 0; i < nb; ++i) {
            const float hk = h[i * H + k];
            float* gi = g + i * 2 * H;
            float* ci = ch + i * H;
            for (int j = 0; j < 2 * H; ++j) gi[j] += hk * wg[j];
            for (int j = 0; j < H; ++j) ci[j] += hk * wc[j];
        }
    }
    for (int i = 0; i < nb; ++i) {
        float* gi = g + i * 2 * H;
        float* ci = ch + i * H;
        float* hi = h + i * H;
        const float* xin = sig[i] >= 0 ? P->Wci + (size_t)sig[i] * H : NULL;
        for (int j = 0; j < 2 * H; ++j) gi[j] = sigmoidf_(gi[j] + P->bg[j]);
        for (int j = 0; j < H; ++j) {
            const float r = gi[j], u = gi[H + j];
            const float xc = (xin ? xin[j] : 0.0f) + P->bci[j];
            const float c = tanhf_(xc + r * (ci[j] + P->bch[j]));
            hi[j] = (1.0f - u) * c + u * hi[j];
        }
        float z0 = 0.0f, z1 = 0.0f;
        for (int j = 0; j < H; ++j) {
            z0 += hi[j] * P->Wd[2 * j];
            z1 += hi[j] * P->Wd[2 * j + 1];
        }
        z0 += P->bd[0];
        z1 += P->bd[1];
        const float m = z0 > z1 ? z0 : z1;
        const float e0 = expf(z0 - m), e1 = expf(z1 - m);
        p[2 * i] = e0 / (e0 + e1);
        p[2 * i + 1] = e1 / (e0 + e1);
    }
}

static gru_params make_params(int H, const float* Wg, const float* bg, const float* Wci, const float* bci,
                              const float* Wch, const float* bch, const float* Wd, const float* bd) {
    gru_params P = {H, Wg, bg, Wci, bci, Wch, bch, Wd, bd};
    return P;
}

/* log P(sigma) of B chains: 1DTFIM/RNNwavefunction.py:76-118 */
int rnnwf_oracle_prnn_log_prob(int H, int N, const float* Wg, const float* bg, const float* Wci, const float* bci,
                               const float* Wch, const float* bch, const float* Wd, const float* bd,
                               const int32_t* samples, int64_t B, double* out, int nthreads) {
    const gru_params P = make_params(H, Wg, bg, Wci, bci, Wch, bch, Wd, bd);
    const int64_t nblk = (B + RB - 1) / RB;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        float* h = (float*)malloc(sizeof(float) * RB * H);
        float* scratch = (float*)malloc(sizeof(float) * RB * 3 * H);
        float p[2 * RB];
        int sig[RB];
        double lp[RB];
#pragma omp for schedule(dynamic, 4)
        for (int64_t b = 0; b < nblk; ++b) {
            const int64_t s0 = b * RB;
            const int nb = (int)((B - s0) < RB ? (B - s0) : RB);
            memset(h, 0, sizeof(float) * RB * H);
            for (int i = 0; i < nb; ++i) { sig[i] = -1; lp[i] = 0.0; }
            for (int n = 0; n < N; ++n) {
                gru_block_step(&P, nb, h, sig, p, scratch);
                for (int i = 0; i < nb; ++i) {
                    const int s = samples[(s0 + i) * N + n];
                    lp[i] += log((double)p[2 * i + s]);     /* probs cast to f64, then log (:113-116) */
                    sig[i] = s;
                }
            }
            for (int i = 0; i < nb; ++i) out[s0 + i] = lp[i];
        }
        free(h);
        free(scratch);
    }
    return 0;
}

/* Ancestral sampling with explicit uniforms u (ns, N): 1DTFIM/RNNwavefunction.py:35-74; the draw
 * follows tf.multinomial's CPU kernel (un-normalised CDF of exp(logit - max) in double). */
int rnnwf_oracle_prnn_sample(int H, int N, const float* Wg, const float* bg, const float* Wci, const float* bci,
                             const float* Wch, const float* bch, const float* Wd, const float* bd,
                             const double* u, int64_t ns, int32_t* samples, double* logp, int nthreads) {
    const gru_params P = make_params(H, Wg, bg, Wci, bci, Wch, bch, Wd, bd);
    const int64_t nblk = (ns + RB - 1) / RB;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        float* h = (float*)malloc(sizeof(float) * RB * H);
        float* scratch = (float*)malloc(sizeof(float) * RB * 3 * H);
        float p[2 * RB];
        int sig[RB];
        double lp[RB];
#pragma omp for schedule(dynamic, 4)
        for (int64_t b = 0; b < nblk; ++b) {
            const int64_t s0 = b * RB;
            const int nb = (int)((ns - s0) < RB ? (ns - s0) : RB);
            memset(h, 0, sizeof(float) * RB * H);
            for (int i = 0; i < nb; ++i) { sig[i] = -1; lp[i] = 0.0; }
            for (int n = 0; n < N; ++n) {
                gru_block_step(&P, nb, h, sig, p, scratch);
                for (int i = 0; i < nb; ++i) {
                    const float l0 = logf(p[2 * i]), l1 = logf(p[2 * i + 1]);
                    const double mx = l0 > l1 ? l0 : l1;
                    const double c0 = exp((double)l0 - mx), c1 = c0 + exp((double)l1 - mx);
                    const double t = u[(s0 + i) * N + n] * c1;
                    const int s = (c0 <= t) ? 1 : 0;
                    samples[(s0 + i) * N + n] = s;
                    lp[i] += log((double)p[2 * i + s]);
                    sig[i] = s;
                }
            }
            if (logp) for (int i = 0; i < nb; ++i) logp[s0 + i] = lp[i];
        }
        free(h);
        free(scratch);
    }
    return 0;
}

/* Ising_local_energies, reference formulation: 1DTFIM/TrainingRNN_1DTFIM.py:13-75.
 * queue: caller scratch (N+1)*ns*N int32; log_probs: caller scratch (N+1)*ns f64 (both filled). */
int rnnwf_oracle_tfim_local_energies(int H, int N, const float* Wg, const float* bg, const float* Wci,
                                     const float* bci, const float* Wch, const float* bch, const float* Wd,
                                     const float* bd, const double* Jz, double Bx, const int32_t* samples,
                                     int64_t ns, int32_t* queue, double* log_probs, double* eloc, int nthreads) {
    for (int64_t s = 0; s < ns; ++s) {                                   /* :31-38 */
        double e = 0.0;
        for (int i = 0; i + 1 < N; ++i)
            e += (samples[s * N + i] == samples[s * N + i + 1] ? 1.0 : -1.0) * (-Jz[i]);
        eloc[s] = e;
    }
    memcpy(queue, samples, sizeof(int32_t) * ns * N);                     /* :40 */
    if (Bx != 0.0) {
        for (int i = 0; i < N; ++i) {                                    /* :43-48 */
            int32_t* q = queue + (size_t)(i + 1) * ns * N;
            memcpy(q, samples, sizeof(int32_t) * ns * N);
            for (int64_t s = 0; s < ns; ++s) q[s * N + i] = 1 - q[s * N + i];
        }
    } else {
        memset(queue + (size_t)ns * N, 0, sizeof(int32_t) * (size_t)N * ns * N);
    }
    const int64_t total = (int64_t)(N + 1) * ns;                         /* :56-65 chunks of <= 25000 */
    const int64_t steps = (total + 24999) / 25000;
    for (int64_t i = 0; i < steps; ++i) {
        const int64_t lo = (i * total) / steps;
        const int64_t hi = i < steps - 1 ? ((i + 1) * total) / steps : total;
        rnnwf_oracle_prnn_log_prob(H, N, Wg, bg, Wci, bci, Wch, bch, Wd, bd, queue + lo * N, hi - lo,
                                   log_probs + lo, nthreads);
    }
    for (int64_t s = 0; s < ns; ++s) {                                   /* :70-74 */
        double acc = 0.0;
        for (int i = 0; i < N; ++i) acc += exp(0.5 * log_probs[(int64_t)(i + 1) * ns + s] - 0.5 * log_probs[s]);
        eloc[s] += -Bx * acc;
    }
    return 0;
}

int rnnwf_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
