// train.hip - device-resident training iteration: the optimizer step and the re-pack of the kernels'
// weight images on the device, and K whole iterations with ONE host synchronisation.
//
// The reference runs an iteration's update as one `sess.run(optstep)` on the device (1DTFIM/TrainingRNN_1DTFIM.py:113,162,221;
// tf.train.AdamOptimizer, beta1 0.9, beta2 0.999, epsilon 1e-8).  Here, per iteration and without a host visit:
//     rnnwf_vmc_step's kernels (samples, local energies, the four moments - left on the device)
//     the gradient's kernels (grad.hip), mean energy and norm read from those moments on the device
//     grad_flat_kernel    dW image -> flat f64 gradient in the order of rnnwf_set_params_flat (table probed from the host unpacker)
//     [RCCL all-reduce of the flat gradient on the stream, multi-rank]
//     adam_kernel         m, v, theta in f64; theta rounded to the model's type as the host optimizer does
//     repack_all_kernel   every weight image rebuilt from theta by replaying the host packers' recorded tables (pack_value.h), one launch
// The arithmetic is the host path's, operation for operation (IEEE f64, no contraction), so a trajectory equals the host-Adam
// one bit for bit (tests/test_gpu_training.py).  Supported: every model of the four drivers - the positive, parity-symmetric, complex
// and float64 GRU with one layer or a stack, and the 2D RNN (rnnwf_device_training_supported answers for a handle).
#include <cmath>

#include "models.h"
#include "pack.h"
#include "pack_split.h"

using namespace rnnwf;

namespace {

constexpr int kMaxSteps = 1024;

__device__ __forceinline__ void repack_entry(const PackEntry& t, const double* __restrict__ P, char* __restrict__ img) {
#pragma clang fp contract(off)
    double v = P[t.a];
    if (t.b >= 0) v = (t.kind & PACK_MINUS) ? __dsub_rn(v, P[t.b]) : __dadd_rn(v, P[t.b]);
    if (t.d >= 0) v = __dadd_rn(v, P[t.d]);
    if (t.c != 1.0) v = __dmul_rn(t.c, v);
    const int kind = t.kind & 255;
    if (kind == PACK_F32) {
        *reinterpret_cast<float*>(img + t.off) = (float)v;
    } else if (kind == PACK_F64) {
        *reinterpret_cast<double*>(img + t.off) = v;
    } else {                                                   // part k of the exact three-way bf16 split of the f32 value (pack_value.h: split3)
        float r = (float)v;
        uint16_t p = 0;
        for (int k = 0; k <= kind - PACK_BF16; ++k) {
            uint32_t u = __float_as_uint(r);
            u += 0x7FFFu + ((u >> 16) & 1u);
            p = (uint16_t)(u >> 16);
            r -= __uint_as_float((uint32_t)p << 16);
        }
        *reinterpret_cast<uint16_t*>(img + t.off) = p;
    }
}

// all images in one launch: entry i belongs to the image whose range of the combined table holds it
struct RepackPlan {
    int nimg;
    int64_t start[9];
    char* target[8];
};
__global__ void repack_all_kernel(const PackEntry* __restrict__ e, RepackPlan plan, const double* __restrict__ P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plan.start[plan.nimg]) return;
    int j = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < plan.nimg && i >= plan.start[k]) j = k;
    repack_entry(e[i], P, plan.target[j]);
}

template <typename T>
__global__ void grad_flat_kernel(const int32_t* __restrict__ sidx, int64_t n, const T* __restrict__ dW, double* __restrict__ G) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t k = sidx[i];
    G[i] = k > 0 ? (double)dW[k - 1] : k < 0 ? -(double)dW[-k - 1] : 0.0;
}

// training.py: Adam.step -  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  theta = (theta - lr_t m / (sqrt(v) + eps)).astype(dtype)
// FUSED (single rank): the flat gradient is gathered from the dW image here (grad_flat_kernel's line) - one launch less per update
template <typename T, bool FUSED>
__global__ void adam_kernel(double* __restrict__ P, double* __restrict__ M, double* __restrict__ V, double* __restrict__ G, int64_t n,
                            double lr_t, double b1, double b2, double eps, int f32, const int32_t* __restrict__ sidx, const T* __restrict__ dW) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (FUSED) {
        const int32_t k = sidx[i];
        G[i] = k > 0 ? (double)dW[k - 1] : k < 0 ? -(double)dW[-k - 1] : 0.0;
    }
    // every operation rounded on its own, as NumPy evaluates the host optimizer's expressions (no fused multiply-add: the
    // translation unit is compiled with -ffp-contract=off, build.py)
    const double g = G[i];
    const double m = __dadd_rn(__dmul_rn(b1, M[i]), __dmul_rn(__dsub_rn(1.0, b1), g));
    const double v = __dadd_rn(__dmul_rn(b2, V[i]), __dmul_rn(__dmul_rn(__dsub_rn(1.0, b2), g), g));
    M[i] = m;
    V[i] = v;
    const double p = __dsub_rn(P[i], __ddiv_rn(__dmul_rn(lr_t, m), __dadd_rn(__dsqrt_rn(v), eps)));
    P[i] = f32 ? (double)(float)p : p;
}

// ---- tables: the host packers once more, over Lin ----------------------------------------------------------------------
template <int NOUT>
int wimg_table(rnnwf_handle* h) {
    switch (h->NFULL) {
        case 1: pack_gru_image<float, 1, NOUT, Lin>(h); return 0;
        case 2: pack_gru_image<float, 2, NOUT, Lin>(h); return 0;
        case 3: pack_gru_image<float, 3, NOUT, Lin>(h); return 0;
        case 4: pack_gru_image<float, 4, NOUT, Lin>(h); return 0;
        case 6: pack_gru_image<float, 6, NOUT, Lin>(h); return 0;
        case 8: pack_gru_image<float, 8, NOUT, Lin>(h); return 0;
        case 12: pack_gru_image<float, 12, NOUT, Lin>(h); return 0;
        case 16: pack_gru_image<float, 16, NOUT, Lin>(h); return 0;
    }
    return 1;
}
// the layouts of split.hip / split_stream.hip (SPLIT_DISPATCH, RLaunch), by width class
template <int NOUT>
int wsplit_table(rnnwf_handle* h) {
    switch (h->NFULL) {
        case 1: pack_split_image<0, 10, NOUT, 1, Lin>(h); return 0;
        case 2: pack_split_image<1, 2, NOUT, 1, Lin>(h); return 0;
        case 3: if (h->H <= 50) pack_split_image<1, 9, NOUT, 2, Lin>(h); else pack_split_image<1, 10, NOUT, 0, Lin>(h); return 0;
        case 4: pack_split_image<2, 2, NOUT, 3, Lin>(h); return 0;
        case 6: pack_split_image<3, 2, NOUT, 3, Lin>(h); return 0;
    }
    return 1;
}
int wimg_table_f64(rnnwf_handle* h) {
    switch (h->NFULL) {
        case 1: pack_gru_image<double, 1, 1, Lin>(h); return 0;
        case 2: pack_gru_image<double, 2, 1, Lin>(h); return 0;
        case 3: pack_gru_image<double, 3, 1, Lin>(h); return 0;
        case 4: pack_gru_image<double, 4, 1, Lin>(h); return 0;
        case 6: pack_gru_image<double, 6, 1, Lin>(h); return 0;
    }
    return 1;
}
int wbasebf_table(rnnwf_handle* h) {
    switch (h->NFULL) {
        case 1: pack_base_bf_image<1, Lin>(h); return 0;
        case 2: pack_base_bf_image<2, Lin>(h); return 0;
        case 3: pack_base_bf_image<3, Lin>(h); return 0;
    }
    return 1;
}

template <typename Fn>
int add_image(rnnwf_handle* h, DevBuf* target, Fn&& run_packer) {
    TrainState& t = h->train;
    if (!target->p) return h->fail(RNNWF_ERR_STATE, "device training: an image the tables rebuild has not been committed");
    PackTable tbl;
    pack_trace().table = &tbl;
    int rc = 0;
    try {
        rc = run_packer();
    } catch (const std::exception& e) {
        pack_trace().table = nullptr;
        return h->fail(RNNWF_ERR_INVALID, "device training: %s", e.what());
    }
    pack_trace().table = nullptr;
    if (rc) return h->fail(RNNWF_ERR_INVALID, "device training: no packer table for this width");
    if (tbl.image_bytes > target->cap) return h->fail(RNNWF_ERR_STATE, "device training: image table larger than its buffer");
    for (const PackEntry& e : tbl.entries) {
        const int kind = e.kind & 255;
        const size_t width = kind == PACK_F32 ? 4 : kind == PACK_F64 ? 8 : 2;
        if (e.a < 0 || e.a >= t.nparams || e.b >= t.nparams || e.d >= t.nparams || kind > PACK_BF16 + 2 || (size_t)e.off + width > tbl.image_bytes)
            return h->fail(RNNWF_ERR_STATE, "device training: image table entry out of range");
    }
    TrainImage& im = t.img[t.nimg++];
    im.n = (int64_t)tbl.entries.size();
    im.target = target;
    if (int r2 = ensure(h, im.table, std::max<size_t>(tbl.entries.size(), 1) * sizeof(PackEntry))) return r2;
    RNNWF_HIP(h, hipMemcpy(im.table.p, tbl.entries.data(), tbl.entries.size() * sizeof(PackEntry), hipMemcpyHostToDevice));
    return 0;
}

int build(rnnwf_handle* h) {
    TrainState& t = h->train;
    if (t.built) return t.supported ? 0 : h->fail(RNNWF_ERR_INVALID, "device-resident training is not available for this model: %s", t.why.c_str());
    t.built = true;
    const bool md = h->model == RNNWF_MODEL_MDRNN2D;
    const bool cplx = h->model == RNNWF_MODEL_CRNN_U1;
    t.nparams = rnnwf_num_params(h);
    for (DevBuf* b : {&t.P, &t.M, &t.V, &t.G})
        if (int rc = ensure(h, *b, (size_t)t.nparams * 8)) return rc;
    RNNWF_HIP(h, hipMemset(t.M.p, 0, (size_t)t.nparams * 8));
    RNNWF_HIP(h, hipMemset(t.V.p, 0, (size_t)t.nparams * 8));
    RNNWF_HIP(h, hipHostMalloc(&t.mom_host, (size_t)kMaxSteps * 4 * sizeof(double), hipHostMallocDefault));
    RNNWF_HIP(h, hipHostGetDevicePointer(&t.mom_host_dev, t.mom_host, 0));
    // gradient: dW image -> flat
    std::vector<int32_t> sidx;
    if (md) {
        t.dw_f64 = true;
        if (int rc = mdrnn_grad_probe(h, sidx, &t.dw_count)) return rc;
    } else if (int rc = grad_flat_probe(h, sidx, &t.dw_count, &t.dw_f64)) return rc;
    if ((int64_t)sidx.size() != t.nparams) return h->fail(RNNWF_ERR_STATE, "device training: gradient probe size mismatch");
    for (int32_t k : sidx)
        if ((size_t)(k < 0 ? -k : k) > t.dw_count) return h->fail(RNNWF_ERR_STATE, "device training: gradient probe index out of range");
    if (int rc = ensure(h, t.gidx, sidx.size() * 4)) return rc;
    RNNWF_HIP(h, hipMemcpy(t.gidx.p, sidx.data(), sidx.size() * 4, hipMemcpyHostToDevice));
    // images (the backward image's table is added on first use: its buffer exists once the gradient has packed it on the host)
    t.nimg = 0;
    if (md) {
        if (int rc = add_image(h, &h->wimg, [&] { return mdrnn_pack_table(h, false); })) return rc;
    } else if (h->NL > 1) {
        // a stack: the forward buffer holds [layer 0 | upper layers] (grad.hip knows the layouts); on the bf16x3 engine (37..50 units)
        // the layer pipeline's images beside it (split.hip: prnn_stack_pack / crnn_stack_pack)
        if (int rc = add_image(h, &h->wimg, [&] { return grad_stack_forward_table(h); })) return rc;
        if (h->engine_split) {
            if (int rc = add_image(h, &h->wsplit, [&] { if (cplx) pack_split_image<1, 9, 3, 2, Lin>(h); else pack_split_image<1, 9, 1, 2, Lin>(h); return 0; })) return rc;
            for (int l = 1; l < h->NL; ++l) {
                const bool top = l == h->NL - 1;
                if (int rc = add_image(h, &h->wsplit_up[l - 1], [&] {
                        if (cplx) pack_split_upper_image<1, 9, 3, Lin>(h, l, top); else pack_split_upper_image<1, 9, 1, Lin>(h, l, top);
                        return 0; })) return rc;
            }
        }
    } else {
        if (int rc = add_image(h, &h->wimg, [&] { return h->f64 ? wimg_table_f64(h) : cplx ? wimg_table<3>(h) : wimg_table<1>(h); })) return rc;
        if (h->engine_split) {
            if (int rc = add_image(h, &h->wsplit, [&] { return cplx ? wsplit_table<3>(h) : wsplit_table<1>(h); })) return rc;
            if (h->NFULL == 6 && !cplx && h->wsplit16.p)
                if (int rc = add_image(h, &h->wsplit16, [&] { pack_split16_image<1, Lin>(h); return 0; })) return rc;
        }
    }
    if (h->base_bf)
        if (int rc = add_image(h, &h->wbasebf, [&] { return wbasebf_table(h); })) return rc;
    t.supported = true;
    return 0;
}

// the backward image's table needs the image's buffer: added on first use, when the gradient has packed it once on the host
int ensure_bwd_image(rnnwf_handle* h) {
    TrainState& t = h->train;
    for (int i = 0; i < t.nimg; ++i)
        if (t.img[i].target == &h->wbwd) return 0;
    if (!h->wbwd.p) return h->fail(RNNWF_ERR_STATE, "device training: the backward image has not been packed yet");
    return add_image(h, &h->wbwd, [&] { return h->model == RNNWF_MODEL_MDRNN2D ? mdrnn_pack_table(h, true) : grad_bwd_pack_table(h); });
}

int params_to_device(rnnwf_handle* h) {
    TrainState& t = h->train;
    if (!t.host_newer) return 0;
    if (int rc = ensure_staging(h, (size_t)t.nparams * 8)) return rc;
    double* flat = (double*)h->staging;
    int64_t off = 0;
    for (auto& kv : h->params) {
        const ParamSpec& p = kv.second;
        for (size_t i = 0; i < p.slot.size(); ++i) flat[off + (int64_t)i] = p.value[(size_t)p.slot[i]];
        off += (int64_t)p.slot.size();
    }
    RNNWF_HIP(h, hipMemcpyAsync(t.P.p, flat, (size_t)t.nparams * 8, hipMemcpyHostToDevice, h->stream));
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    t.host_newer = false;
    t.dev_newer = false;
    return 0;
}

// the tables of all images back to back (rebuilt when an image joins: the backward image's table is added on first use)
int ensure_combo(rnnwf_handle* h) {
    TrainState& t = h->train;
    if (t.combo_nimg == t.nimg) return 0;
    int64_t total = 0;
    for (int i = 0; i < t.nimg; ++i) total += t.img[i].n;
    if (int rc = ensure(h, t.combo, std::max<int64_t>(total, 1) * sizeof(PackEntry))) return rc;
    int64_t off = 0;
    for (int i = 0; i < t.nimg; ++i) {
        if (t.img[i].n)
            RNNWF_HIP(h, hipMemcpyAsync((char*)t.combo.p + off * sizeof(PackEntry), t.img[i].table.p, t.img[i].n * sizeof(PackEntry),
                                        hipMemcpyDeviceToDevice, h->stream));
        off += t.img[i].n;
    }
    t.combo_nimg = t.nimg;
    return 0;
}

int launch_update(rnnwf_handle* h, double lr_t, double b1, double b2, double eps) {
    TrainState& t = h->train;
    const unsigned blocks = (unsigned)((t.nparams + 255) / 256);
    const int f32 = h->f64 ? 0 : 1;
    if (h->comm) {            // (also with a one-rank communicator: the single-GPU test walks the road)
        if (t.dw_f64) grad_flat_kernel<double><<<blocks, 256, 0, h->stream>>>((const int32_t*)t.gidx.p, t.nparams, (const double*)h->gradW.p, (double*)t.G.p);
        else grad_flat_kernel<float><<<blocks, 256, 0, h->stream>>>((const int32_t*)t.gidx.p, t.nparams, (const float*)h->gradW.p, (double*)t.G.p);
        RNNWF_HIP(h, hipGetLastError());
        if (int rc = comm_allreduce_device(h, t.G.p, (size_t)t.nparams)) return rc;       // one in-stream RCCL sum of the gradient
        adam_kernel<float, false><<<blocks, 256, 0, h->stream>>>((double*)t.P.p, (double*)t.M.p, (double*)t.V.p, (double*)t.G.p, t.nparams, lr_t, b1, b2,
                                                                 eps, f32, nullptr, nullptr);
    } else if (t.dw_f64) {
        adam_kernel<double, true><<<blocks, 256, 0, h->stream>>>((double*)t.P.p, (double*)t.M.p, (double*)t.V.p, (double*)t.G.p, t.nparams, lr_t, b1, b2,
                                                                 eps, f32, (const int32_t*)t.gidx.p, (const double*)h->gradW.p);
    } else {
        adam_kernel<float, true><<<blocks, 256, 0, h->stream>>>((double*)t.P.p, (double*)t.M.p, (double*)t.V.p, (double*)t.G.p, t.nparams, lr_t, b1, b2,
                                                                eps, f32, (const int32_t*)t.gidx.p, (const float*)h->gradW.p);
    }
    RNNWF_HIP(h, hipGetLastError());
    if (int rc = ensure_combo(h)) return rc;
    RepackPlan plan{};
    plan.nimg = t.nimg;
    for (int i = 0; i < t.nimg; ++i) {
        plan.start[i + 1] = plan.start[i] + t.img[i].n;
        plan.target[i] = (char*)t.img[i].target->p;
    }
    if (const int64_t total = plan.start[t.nimg]) {
        repack_all_kernel<<<(unsigned)((total + 255) / 256), 256, 0, h->stream>>>((const PackEntry*)t.combo.p, plan, (const double*)t.P.p);
        RNNWF_HIP(h, hipGetLastError());
    }
    t.dev_newer = true;
    return 0;
}

double adam_lr_t(double lr, double b1, double b2, int64_t t) {
    return lr * std::sqrt(1.0 - std::pow(b2, (double)t)) / (1.0 - std::pow(b1, (double)t));
}

}  // namespace

// rnnwf_allreduce_grads for the models with a device-side flat gradient: dW image -> flat f64 on the device -> ONE in-stream RCCL sum ->
// host arrays (no device -> host map -> pinned staging -> device round trip in front of the collective).  Returns 1 when it handled
// the call, 0 when the caller must take the staged road, < 0 on error.
int rnnwf::train_allreduce_grads_device(rnnwf_handle* h) {
    TrainState& t = h->train;
    if (!h->comm || !h->gradW.p) return 0;
    if (!t.built) { if (build(h) != 0) { h->err.clear(); return 0; } }
    if (!t.supported) return 0;
    const unsigned blocks = (unsigned)((t.nparams + 255) / 256);
    if (t.dw_f64) grad_flat_kernel<double><<<blocks, 256, 0, h->stream>>>((const int32_t*)t.gidx.p, t.nparams, (const double*)h->gradW.p, (double*)t.G.p);
    else grad_flat_kernel<float><<<blocks, 256, 0, h->stream>>>((const int32_t*)t.gidx.p, t.nparams, (const float*)h->gradW.p, (double*)t.G.p);
    if (hipGetLastError() != hipSuccess) return h->fail(RNNWF_ERR_HIP, "grad_flat_kernel launch failed");
    if (int rc = comm_allreduce_device(h, t.G.p, (size_t)t.nparams)) return rc < 0 ? rc : -1;
    if (int rc = ensure_staging(h, (size_t)t.nparams * 8)) return rc < 0 ? rc : -1;
    if (hipMemcpyAsync(h->staging, t.G.p, (size_t)t.nparams * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess)
        return h->fail(RNNWF_ERR_HIP, "gradient download failed");
    const double* flat = (const double*)h->staging;
    int64_t off = 0;
    for (auto& kv : h->params) {
        std::vector<double>& g = h->grads[kv.first];
        g.assign(kv.second.value.size(), 0.0);
        for (size_t i = 0; i < kv.second.slot.size(); ++i) g[(size_t)kv.second.slot[i]] = flat[off + (int64_t)i];
        off += (int64_t)kv.second.slot.size();
    }
    return 1;
}

void rnnwf::train_params_changed_on_host(rnnwf_handle* h) {
    h->train.host_newer = true;
    h->train.dev_newer = false;
}

int rnnwf::train_sync_params_to_host(rnnwf_handle* h) {
    TrainState& t = h->train;
    if (!t.supported || !t.dev_newer) return 0;
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = ensure_staging(h, (size_t)t.nparams * 8)) return rc;
    RNNWF_HIP(h, hipMemcpyAsync(h->staging, t.P.p, (size_t)t.nparams * 8, hipMemcpyDeviceToHost, h->stream));
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    const double* flat = (const double*)h->staging;
    int64_t off = 0;
    for (auto& kv : h->params) {
        ParamSpec& p = kv.second;
        for (size_t i = 0; i < p.slot.size(); ++i) p.value[(size_t)p.slot[i]] = flat[off + (int64_t)i];
        off += (int64_t)p.slot.size();
    }
    t.dev_newer = false;
    return 0;
}

extern "C" int rnnwf_device_training_supported(rnnwf_handle* h) {
    if (!h || !h->committed) return 0;
    if (hipSetDevice(h->cfg.device) != hipSuccess) return 0;
    return build(h) == 0 ? 1 : 0;
}

extern "C" int rnnwf_adam_set_state(rnnwf_handle* h, const double* m_flat, const double* v_flat, int64_t count, int64_t t_steps) {
    if (!h || !h->committed) return RNNWF_ERR_INVALID;
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = build(h)) return rc;
    TrainState& t = h->train;
    if (count != t.nparams || t_steps < 0) return h->fail(RNNWF_ERR_INVALID, "rnnwf_adam_set_state: the model has %lld parameters, caller passed %lld", (long long)t.nparams, (long long)count);
    if (m_flat && v_flat) {
        RNNWF_HIP(h, hipMemcpy(t.M.p, m_flat, (size_t)count * 8, hipMemcpyHostToDevice));
        RNNWF_HIP(h, hipMemcpy(t.V.p, v_flat, (size_t)count * 8, hipMemcpyHostToDevice));
    } else {
        RNNWF_HIP(h, hipMemset(t.M.p, 0, (size_t)count * 8));
        RNNWF_HIP(h, hipMemset(t.V.p, 0, (size_t)count * 8));
    }
    t.adam_t = t_steps;
    return RNNWF_OK;
}

extern "C" int rnnwf_adam_get_state(rnnwf_handle* h, double* m_flat, double* v_flat, int64_t count, int64_t* t_steps) {
    if (!h || !h->committed) return RNNWF_ERR_INVALID;
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = build(h)) return rc;
    TrainState& t = h->train;
    if (count != t.nparams) return h->fail(RNNWF_ERR_INVALID, "rnnwf_adam_get_state: the model has %lld parameters, caller passed %lld", (long long)t.nparams, (long long)count);
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    if (m_flat) RNNWF_HIP(h, hipMemcpy(m_flat, t.M.p, (size_t)count * 8, hipMemcpyDeviceToHost));
    if (v_flat) RNNWF_HIP(h, hipMemcpy(v_flat, t.V.p, (size_t)count * 8, hipMemcpyDeviceToHost));
    if (t_steps) *t_steps = t.adam_t;
    return RNNWF_OK;
}

// One optimizer step from the gradient rnnwf_vmc_gradient left on the device (its dW image), then the images' re-pack: what
// `sess.run(optstep)` does behind the gradient (1DTFIM/TrainingRNN_1DTFIM.py:221).
extern "C" int rnnwf_adam_step(rnnwf_handle* h, double learning_rate, double beta1, double beta2, double epsilon) {
    if (!h || !h->committed) return RNNWF_ERR_INVALID;
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = build(h)) return rc;
    TrainState& t = h->train;
    if (!h->gradW.p || h->grads.empty()) return h->fail(RNNWF_ERR_STATE, "rnnwf_adam_step: no gradient (call rnnwf_vmc_gradient first)");
    if (int rc = params_to_device(h)) return rc;
    if (int rc = ensure_bwd_image(h)) return rc;
    t.adam_t += 1;
    if (int rc = launch_update(h, adam_lr_t(learning_rate, beta1, beta2, t.adam_t), beta1, beta2, epsilon)) return rc;
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    h->last_ns = 0;                       // the resident batch belongs to the old weights
    return RNNWF_OK;
}

// K iterations (K <= 1024), one host synchronisation; moments: [K][4] = {sum Re E, sum (Re E)^2, n, sum Im E} of every iteration's
// batch (summed over the ranks when the in-step all-reduce is on).  Iteration k draws with step index step0 + k and uses
// learning_rates[k] (the 2D drivers adapt it per iteration: Training2DRNN_2DTFIM.py:228).
extern "C" int rnnwf_train_steps(rnnwf_handle* h, int32_t K, int64_t numsamples, uint64_t seed, uint64_t step0, int64_t sample_offset,
                                 const double* couplings, int64_t n_couplings, const double* learning_rates, double beta1, double beta2,
                                 double epsilon, double* moments) {
    if (!h || !h->committed) return RNNWF_ERR_INVALID;
    if (K < 1 || K > kMaxSteps || numsamples < 1 || !couplings || !learning_rates || !moments)
        return h->fail(RNNWF_ERR_INVALID, "rnnwf_train_steps: bad arguments (1 <= K <= %d)", kMaxSteps);
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = build(h)) return rc;
    TrainState& t = h->train;
    const bool cplx = h->model == RNNWF_MODEL_CRNN_U1, md = h->model == RNNWF_MODEL_MDRNN2D;
    if (n_couplings != (cplx ? 3 * h->N + 2 : h->N + 1)) return h->fail(RNNWF_ERR_INVALID, "rnnwf_train_steps: wrong number of couplings");
    if (int rc = params_to_device(h)) return rc;
    for (int k = 0; k < K; ++k) {
        int rc;
        // single rank: the moments kernel writes iteration k's row of the pinned table itself (no copy launch per iteration)
        h->moments_direct = h->reduce_in_step ? nullptr : (double*)t.mom_host_dev + 4 * k;
        if (cplx) rc = crnn_vmc_step(h, numsamples, seed, step0 + (uint64_t)k, sample_offset, couplings, nullptr, nullptr, nullptr);
        else if (md) rc = mdrnn_vmc_step(h, numsamples, seed, step0 + (uint64_t)k, sample_offset, couplings, nullptr, nullptr, nullptr);
        else rc = prnn_vmc_step(h, numsamples, seed, step0 + (uint64_t)k, sample_offset, couplings, nullptr, nullptr, nullptr);
        h->moments_direct = nullptr;
        if (rc) return rc;
        if (h->reduce_in_step)
            RNNWF_HIP(h, hipMemcpyAsync((double*)t.mom_host + 4 * k, h->moments.p, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (int r2 = md ? mdrnn_grad_device(h, 0.0, 0.0, (const double*)h->moments.p, nullptr)
                        : grad_single_layer_device(h, 0.0, 0.0, 0.0, (const double*)h->moments.p, nullptr)) return r2;
        if (int r2 = ensure_bwd_image(h)) return r2;
        if (int r2 = launch_update(h, adam_lr_t(learning_rates[k], beta1, beta2, t.adam_t + k + 1), beta1, beta2, epsilon)) return r2;
    }
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    memcpy(moments, t.mom_host, (size_t)K * 4 * sizeof(double));
    t.adam_t += K;
    h->last_ns = 0;                       // the resident batch belongs to the weights before the last update
    h->grads.clear();
    return RNNWF_OK;
}
