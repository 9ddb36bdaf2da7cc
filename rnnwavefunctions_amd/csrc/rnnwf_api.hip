// rnnwf_api.hip - C ABI (include/rnnwf.h) over the gfx950 kernels: handle life cycle, parameters,
// pRNN sample / log_prob / TFIM local energies, fused VMC step, timing.
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>

#include "handle.h"
#include "init_params.h"
#include "models.h"
#include "pack.h"
#include "util_kernels.h"

using namespace rnnwf;

static std::string g_create_error;

// -------------------------------------------------------------------------------------------------
// parameter table per model
// -------------------------------------------------------------------------------------------------
// rows / cols: segments (true length, padded length) of the two axes (a vector has one row segment {1, 1}); the caller's element
// (i, j) lands at the padded position of its segment
typedef std::vector<std::pair<int64_t, int64_t>> Segs;
static void declare(rnnwf_handle* h, const std::string& name, std::vector<int64_t> shape, const Segs& rows, const Segs& cols) {
    ParamSpec s;
    s.shape = shape;
    int64_t tr = 0, tc = 0, pr = 0, pc = 0;
    for (auto& g : rows) { tr += g.first; pr += g.second; }
    for (auto& g : cols) { tc += g.first; pc += g.second; }
    s.value.assign((size_t)(pr * pc), 0.0);
    std::vector<int64_t> rmap, cmap;
    { int64_t off = 0; for (auto& g : rows) { for (int64_t i = 0; i < g.first; ++i) rmap.push_back(off + i); off += g.second; } }
    { int64_t off = 0; for (auto& g : cols) { for (int64_t j = 0; j < g.first; ++j) cmap.push_back(off + j); off += g.second; } }
    s.slot.reserve((size_t)(tr * tc));
    for (int64_t i = 0; i < tr; ++i)
        for (int64_t j = 0; j < tc; ++j) s.slot.push_back(rmap[i] * pc + cmap[j]);
    h->params[name] = s;
}
static void declare(rnnwf_handle* h, const std::string& name, std::vector<int64_t> shape) {      // nothing padded
    if (shape.size() == 1) declare(h, name, shape, {{1, 1}}, {{shape[0], shape[0]}});
    else declare(h, name, shape, {{shape[0], shape[0]}}, {{shape[1], shape[1]}});
}

static void declare_params(rnnwf_handle* h) {
    const int64_t H = h->H;
    if (h->model == RNNWF_MODEL_MDRNN2D) {
        declare(h, "Wh_rnn_0", {H, H});
        declare(h, "Uh_rnn_0", {2, H});
        declare(h, "Wv_rnn_0", {H, H});
        declare(h, "Uv_rnn_0", {2, H});
        declare(h, "b_rnn_0", {H});
        declare(h, "wf_dense/kernel", {H, 2});
        declare(h, "wf_dense/bias", {2});
        return;
    }
    // layer l of width w_l (input width d_l: 2 for the first layer, w_{l-1} above), everything padded to H = max w
    for (int l = 0; l < h->NL; ++l) {
        const std::string pl = "multi_rnn_cell/cell_" + std::to_string(l) + "/cudnn_compatible_gru_cell/";
        const int64_t w = h->cfg.units[l], d = l ? h->cfg.units[l - 1] : 2, dp = l ? H : 2;
        declare(h, pl + "gates/kernel", {d + w, 2 * w}, {{d, dp}, {w, H}}, {{w, H}, {w, H}});
        declare(h, pl + "gates/bias", {2 * w}, {{1, 1}}, {{w, H}, {w, H}});
        declare(h, pl + "candidate/input_projection/kernel", {d, w}, {{d, dp}}, {{w, H}});
        declare(h, pl + "candidate/input_projection/bias", {w}, {{1, 1}}, {{w, H}});
        declare(h, pl + "candidate/hidden_projection/kernel", {w, w}, {{w, H}}, {{w, H}});
        declare(h, pl + "candidate/hidden_projection/bias", {w}, {{1, 1}}, {{w, H}});
    }
    const int64_t wt = h->cfg.units[h->NL - 1];
    if (h->model == RNNWF_MODEL_CRNN_U1) {
        declare(h, "wf_dense_ampl/kernel", {wt, 2}, {{wt, H}}, {{2, 2}});
        declare(h, "wf_dense_ampl/bias", {2});
        declare(h, "wf_dense_phase/kernel", {wt, 2}, {{wt, H}}, {{2, 2}});
        declare(h, "wf_dense_phase/bias", {2});
    } else {
        declare(h, "wf_dense/kernel", {wt, 2}, {{wt, H}}, {{2, 2}});
        declare(h, "wf_dense/bias", {2});
    }
}

size_t rnnwf::state_budget_bytes(const rnnwf_handle* h, size_t dflt) {
    return h->knobs.state_budget ? h->knobs.state_budget : dflt;
}

// The environment is consulted here and nowhere else: once per handle, at creation.
// Returns nullptr, or the offending setting: a value this build does not know is refused (rnnwf_create fails), never ignored.
static const char* read_knobs(Knobs& k) {
    if (const char* e = getenv("RNNWF_ENGINE")) {
        k.engine = !strcmp(e, "f32") ? 1 : !strcmp(e, "bf16x3") ? 2 : -1;
#ifdef RNNWF_DIAGNOSTICS      // measured negatives kept for A/B runs (docs/history): the release library holds one kernel per model and width class
        if (const char* e2 = getenv("RNNWF_MDRNN_PREFETCH")) k.md_prefetch = !strcmp(e2, "1");
        if (k.engine < 0) k.engine = !strcmp(e, "bf16x3-serial") ? 3 : !strcmp(e, "bf16x3-hipcc") ? 4 : !strcmp(e, "bf16x3-asm32") ? 5 : !strcmp(e, "bf16x3-n16") ? 7 : -1;
#endif
        if (k.engine < 0 && *e) return "RNNWF_ENGINE (this build knows: f32, bf16x3)";
        if (k.engine < 0) k.engine = 0;
    }
#ifdef RNNWF_DIAGNOSTICS
    if (const char* e = getenv("RNNWF_MDRNN_PREFETCH")) k.md_prefetch = !strcmp(e, "1");
#endif
    if (const char* e = getenv("RNNWF_NO_COOP")) k.no_coop = atoi(e) != 0;
    if (const char* e = getenv("RNNWF_BASE")) k.base_f32 = !strcmp(e, "f32");
    if (const char* e = getenv("RNNWF_STATE_BUDGET_MB")) {
        const long long mb = atoll(e);
        if (mb > 0) k.state_budget = (size_t)mb << 20;
    }
#ifdef RNNWF_DIAGNOSTICS      // timing-only switches that produce WRONG numbers: tools/ builds only (rnnwf_backend_name says so)
    if (const char* e = getenv("RNNWF_ABLATE")) k.ablate = atoi(e);
    if (const char* e = getenv("RNNWF_ABLATE_BASE")) k.ablate_base = atoi(e);
#endif
    return nullptr;
}

int rnnwf::upload_couplings(rnnwf_handle* h, const double* src, size_t n) {
    if (h->coupl.p && h->coupl_host.size() == n && std::equal(src, src + n, h->coupl_host.begin())) return 0;
    if (int rc = ensure(h, h->coupl, n * 8)) return rc;
    h->coupl_host.assign(src, src + n);
    RNNWF_HIP(h, hipMemcpyAsync(h->coupl.p, h->coupl_host.data(), n * 8, hipMemcpyHostToDevice, h->stream));
    return 0;
}

static int pick_nfull(int H, bool f64, bool mdrnn) {
    const int cand[] = {1, 2, 3, 4, 5, 6, 8, 12, 16};
    for (int nf : cand) {
        if (nf == 5 && !mdrnn) continue;          // (the 84-unit layout exists for the 2D RNN only)
        if (16 * nf + 4 < H) continue;
        if (mdrnn && nf > 5) return -1;           // the 2D RNN's image must fit the 160 KiB LDS: <= 84 units
        if (f64 && nf > 6) return -1;             // float64 GRU: <= 100 units (above 68 the image is read through L2)
        return nf;                                // float GRU above 100 units (nf 8, 12, 16: <= 132, 196, 260): image read through L2
    }
    return -1;
}

// -------------------------------------------------------------------------------------------------
// life cycle
// -------------------------------------------------------------------------------------------------
#ifdef RNNWF_DIAGNOSTICS
extern "C" const char* rnnwf_backend_name(void) { return "hip-gfx950-diagnostics"; }
#else
extern "C" const char* rnnwf_backend_name(void) { return "hip-gfx950"; }
#endif
extern "C" int rnnwf_abi_version(void) { return RNNWF_ABI_VERSION; }

extern "C" const char* rnnwf_last_error(const rnnwf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int rnnwf_create(const rnnwf_config* cfg, rnnwf_handle** out) {
    auto bad = [&](const std::string& m) {
        g_create_error = m;
        return (int)RNNWF_ERR_INVALID;
    };
    if (!cfg || !out) return bad("rnnwf_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != RNNWF_ABI_VERSION) return bad("rnnwf_create: ABI version mismatch");
    if (cfg->model < 0 || cfg->model > RNNWF_MODEL_MDRNN2D) return bad("rnnwf_create: unknown model");
    if (cfg->nx < 1 || cfg->ny < 1) return bad("rnnwf_create: system size must be positive");
    if (cfg->num_layers < 1 || cfg->num_layers > RNNWF_MAX_LAYERS)
        return bad("rnnwf_create: len(units) must be 1.." + std::to_string(RNNWF_MAX_LAYERS));
    if (cfg->units[0] < 1) return bad("rnnwf_create: units[0] must be positive");
    if (cfg->num_layers > 1) {
        // MultiRNNCell stacks (1DTFIM/RNNwavefunction.py:32, J1J2/ComplexRNNwavefunction.py:40,
        // 2DTFIM_1DRNN/RNNwavefunction.py): GRU models, equal widths, the images of all layers must fit LDS
        if (cfg->model == RNNWF_MODEL_MDRNN2D)
            return bad("rnnwf_create: the 2D RNN has one layer (the reference: 'num_layers is not supported yet', 2DTFIM_2DRNN/run_2dTFIM.py:9)");
        for (int l = 1; l < cfg->num_layers; ++l)
            if (cfg->units[l] < 1) return bad("rnnwf_create: every units[n] must be positive");
        // two layer images (one forward + its two backward operands in the gradient pass) must fit the 160 KB of LDS;
        // a third layer's image is read through L2 where three do not fit (layout.h: MlSpill)
        if (cfg->model == RNNWF_MODEL_GRU1D_F64) {
            if (*std::max_element(cfg->units, cfg->units + cfg->num_layers) > 68)
                return bad("rnnwf_create: stacked float64 layers: num_units <= 68");
        } else if (*std::max_element(cfg->units, cfg->units + cfg->num_layers) > 100) {
            // (above 52 units the upper layers' images are read through L2: layout.h, MlSpill)
            return bad("rnnwf_create: stacked layers: num_units <= 100");
        }
    }
    const bool two_d = cfg->model == RNNWF_MODEL_MDRNN2D || cfg->model == RNNWF_MODEL_GRU1D_F64;
    if (!two_d && cfg->ny != 1) return bad("rnnwf_create: ny must be 1 for the 1D models");
    if (cfg->model == RNNWF_MODEL_CRNN_U1 && (cfg->nx % 2)) return bad("rnnwf_create: the U(1) cRNN needs an even number of sites");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_error = std::string("rnnwf_create: no HIP device available (") + hipGetErrorString(e) + ")";
        return RNNWF_ERR_HIP;
    }
    if (cfg->device < 0 || cfg->device >= ndev) return bad("rnnwf_create: device ordinal out of range");

    rnnwf_handle* h = new rnnwf_handle();
    h->cfg = *cfg;
    h->model = cfg->model;
    h->f64 = cfg->model == RNNWF_MODEL_GRU1D_F64 || cfg->model == RNNWF_MODEL_MDRNN2D;
    h->H = *std::max_element(cfg->units, cfg->units + cfg->num_layers);      // layers of unequal width are padded to the widest (handle.h: ParamSpec)
    h->Nx = cfg->nx;
    h->Ny = cfg->ny;
    h->N = cfg->nx * cfg->ny;
    h->NL = cfg->num_layers;
    h->NFULL = pick_nfull(h->H, h->f64, h->model == RNNWF_MODEL_MDRNN2D);
    if (h->NFULL < 0) {
        delete h;
        return bad("rnnwf_create: num_units too large (f32 GRU: <= 260, f64 GRU: <= 100, 2D RNN: <= 84)");
    }
    auto fail_hip = [&](const char* what, hipError_t err) {
        g_create_error = std::string("rnnwf_create: ") + what + ": " + hipGetErrorString(err);
        delete h;
        return (int)RNNWF_ERR_HIP;
    };
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) return fail_hip("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, cfg->device)) != hipSuccess) return fail_hip("hipGetDeviceProperties", e);
    h->cu_count = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return fail_hip("hipStreamCreate", e);
    if ((e = hipHostMalloc(&h->pinned, 4096, hipHostMallocDefault)) != hipSuccess) return fail_hip("hipHostMalloc", e);
    if ((e = hipHostGetDevicePointer(&h->pinned_dev, h->pinned, 0)) != hipSuccess) return fail_hip("hipHostGetDevicePointer", e);
    if (const char* badknob = read_knobs(h->knobs)) {
        g_create_error = std::string("rnnwf_create: unknown value of the environment switch ") + badknob;
        rnnwf_destroy(h);
        return RNNWF_ERR_INVALID;
    }
    declare_params(h);
    {   // padded index -> flat parameter index (the order of rnnwf_set_params_flat: tensors by name, each in the caller's shape)
        int32_t off = 0;
        for (auto& kv : h->params) {
            std::vector<int32_t>& f = h->param_flat[kv.first];
            f.assign(kv.second.value.size(), -1);
            for (size_t i = 0; i < kv.second.slot.size(); ++i) f[(size_t)kv.second.slot[i]] = off + (int32_t)i;
            off += (int32_t)kv.second.slot.size();
        }
    }
    *out = h;
    return RNNWF_OK;
}

static void free_buf(DevBuf& b) {
    if (b.p) hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

extern "C" int rnnwf_destroy(rnnwf_handle* h) {
    if (!h) return RNNWF_OK;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    rnnwf_comm_destroy(h);
    DevBuf* bufs[] = {&h->wimg, &h->samples_i32, &h->bits, &h->bits2, &h->hck, &h->lpq, &h->lpq2, &h->out_lp,
                      &h->out_lp2, &h->eloc, &h->moments, &h->coupl, &h->maps, &h->camp, &h->tiles,
                      &h->tile_count, &h->cbase, &h->cout, &h->rowbuf, &h->wbwd, &h->gradP, &h->gradQ, &h->gradW, &h->gradPart, &h->gradHeadPart, &h->wsplit, &h->wsplit16, &h->wbasebf, &h->gradDX[0], &h->gradDX[1], &h->reduce_scratch,
                      &h->xrec[0], &h->xrec[1], &h->wsplit_up[0], &h->wsplit_up[1], &h->wsplit_up[2],
                      &h->train.P, &h->train.M, &h->train.V, &h->train.G, &h->train.gidx, &h->train.img[0].table, &h->train.img[1].table,
                      &h->train.img[2].table, &h->train.img[3].table, &h->train.img[4].table, &h->train.img[5].table, &h->train.img[6].table,
                      &h->train.img[7].table, &h->train.combo};
    static_assert(RNNWF_MAX_LAYERS == 4, "wsplit_up has RNNWF_MAX_LAYERS - 1 entries");
    for (DevBuf* b : bufs) free_buf(*b);
    for (auto& t : h->timers) {
        for (auto& ev : t.pending) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
        for (auto& ev : t.pool) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    }
    if (h->pinned) hipHostFree(h->pinned);
    if (h->train.mom_host) hipHostFree(h->train.mom_host);
    if (h->staging) hipHostFree(h->staging);
    if (h->upbuf) hipHostFree(h->upbuf);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return RNNWF_OK;
}

// -------------------------------------------------------------------------------------------------
// parameters
// -------------------------------------------------------------------------------------------------
static int find_param(rnnwf_handle* h, const char* name, int64_t count, ParamSpec** out) {
    if (!name) return h->fail(RNNWF_ERR_INVALID, "parameter name is null");
    auto it = h->params.find(name);
    if (it == h->params.end()) return h->fail(RNNWF_ERR_INVALID, "unknown parameter '%s' for this model", name);
    if ((int64_t)it->second.slot.size() != count)
        return h->fail(RNNWF_ERR_INVALID, "parameter '%s' has %lld elements, caller passed %lld", name,
                       (long long)it->second.slot.size(), (long long)count);
    *out = &it->second;
    return 0;
}

extern "C" int rnnwf_set_param(rnnwf_handle* h, const char* name, const void* data, int64_t count, int32_t dtype) {
    if (!h || !data) return RNNWF_ERR_INVALID;
    if (int rc = train_sync_params_to_host(h)) return rc;         // (the other tensors must be current before this one is replaced)
    train_params_changed_on_host(h);
    ParamSpec* p;
    if (int rc = find_param(h, name, count, &p)) return rc;
    if (dtype == RNNWF_F32) for (int64_t i = 0; i < count; ++i) p->value[p->slot[i]] = ((const float*)data)[i];
    else if (dtype == RNNWF_F64) for (int64_t i = 0; i < count; ++i) p->value[p->slot[i]] = ((const double*)data)[i];
    else return h->fail(RNNWF_ERR_INVALID, "unknown dtype %d", dtype);
    p->set = true;
    h->committed = false;
    return RNNWF_OK;
}

extern "C" int rnnwf_get_param(rnnwf_handle* h, const char* name, void* data, int64_t count, int32_t dtype) {
    if (!h || !data) return RNNWF_ERR_INVALID;
    if (int rc = train_sync_params_to_host(h)) return rc;         // device-resident optimizer steps the host copy has not seen
    ParamSpec* p;
    if (int rc = find_param(h, name, count, &p)) return rc;
    if (dtype == RNNWF_F32) for (int64_t i = 0; i < count; ++i) ((float*)data)[i] = (float)p->value[p->slot[i]];
    else if (dtype == RNNWF_F64) for (int64_t i = 0; i < count; ++i) ((double*)data)[i] = p->value[p->slot[i]];
    else return h->fail(RNNWF_ERR_INVALID, "unknown dtype %d", dtype);
    return RNNWF_OK;
}

// All parameters in ONE call: `flat` holds every tensor in the caller's shapes, concatenated in the order of their names
// (byte-wise sorted, the order of rnnwf_param_name); commits.  What a training loop calls once per iteration instead of
// 8..26 rnnwf_set_param calls + rnnwf_commit_params.
extern "C" int rnnwf_set_params_flat(rnnwf_handle* h, const double* flat, int64_t count) {
    if (!h || !flat) return RNNWF_ERR_INVALID;
    if (count != rnnwf_num_params(h))
        return h->fail(RNNWF_ERR_INVALID, "rnnwf_set_params_flat: the model has %lld parameters, caller passed %lld",
                       (long long)rnnwf_num_params(h), (long long)count);
    int64_t off = 0;
    for (auto& kv : h->params) {
        ParamSpec& p = kv.second;
        for (size_t i = 0; i < p.slot.size(); ++i) p.value[p.slot[i]] = flat[off + (int64_t)i];
        off += (int64_t)p.slot.size();
        p.set = true;
    }
    train_params_changed_on_host(h);
    h->committed = false;
    return rnnwf_commit_params(h);
}
// name of the i-th tensor of that order (nullptr past the end) and its element count
extern "C" const char* rnnwf_param_name(const rnnwf_handle* h, int32_t i, int64_t* count) {
    if (!h || i < 0) return nullptr;
    for (auto& kv : h->params)
        if (i-- == 0) {
            if (count) *count = (int64_t)kv.second.slot.size();
            return kv.first.c_str();
        }
    return nullptr;
}

extern "C" int64_t rnnwf_num_params(const rnnwf_handle* h) {
    int64_t n = 0;
    if (h) for (auto& kv : h->params) n += (int64_t)kv.second.slot.size();
    return n;
}

extern "C" int rnnwf_init_params(rnnwf_handle* h, uint64_t seed) {
    if (!h) return RNNWF_ERR_INVALID;
    train_params_changed_on_host(h);
    if (seed > 0xffffffffull) return h->fail(RNNWF_ERR_INVALID, "rnnwf_init_params: seed must fit 32 bits (numpy.random.RandomState)");
    NumpyRandomState rng((uint32_t)seed);
    const int64_t H = h->H;
    const bool f32 = !h->f64;
    auto draw = [&](const std::string& name, int64_t rows, int64_t cols, bool vector = false) {
        auto& p = h->params.at(name);
        std::vector<double> tmp;                               // the caller's shape (what the numpy stream is drawn for), then padded
        glorot_fill(rng, rows, cols, vector, f32, tmp);
        for (size_t i = 0; i < p.slot.size(); ++i) p.value[p.slot[i]] = tmp[i];
        p.set = true;
    };
    auto constant = [&](const std::string& name, double v) {
        auto& p = h->params.at(name);
        for (size_t i = 0; i < p.slot.size(); ++i) p.value[p.slot[i]] = v;          // padded entries stay 0
        p.set = true;
    };
    if (h->model == RNNWF_MODEL_MDRNN2D) {                 // params.init_mdrnn_params: every tensor xavier, incl. b
        draw("Wh_rnn_0", H, H);
        draw("Uh_rnn_0", 2, H);
        draw("Wv_rnn_0", H, H);
        draw("Uv_rnn_0", 2, H);
        draw("b_rnn_0", 1, H, true);
        draw("wf_dense/kernel", H, 2);
        constant("wf_dense/bias", 0.0);
    } else {                                               // params.init_gru_params
        int64_t d = 2, w = H;
        for (int l = 0; l < h->NL; ++l) {
            const std::string pre = "multi_rnn_cell/cell_" + std::to_string(l) + "/cudnn_compatible_gru_cell/";
            w = h->cfg.units[l];
            draw(pre + "gates/kernel", d + w, 2 * w);
            constant(pre + "gates/bias", 1.0);
            draw(pre + "candidate/input_projection/kernel", d, w);
            constant(pre + "candidate/input_projection/bias", 0.0);
            draw(pre + "candidate/hidden_projection/kernel", w, w);
            constant(pre + "candidate/hidden_projection/bias", 0.0);
            d = w;
        }
        if (h->model == RNNWF_MODEL_CRNN_U1) {
            draw("wf_dense_ampl/kernel", w, 2);
            constant("wf_dense_ampl/bias", 0.0);
            draw("wf_dense_phase/kernel", w, 2);
            constant("wf_dense_phase/bias", 0.0);
        } else {
            draw("wf_dense/kernel", w, 2);
            constant("wf_dense/bias", 0.0);
        }
    }
    h->committed = false;
    return rnnwf_commit_params(h);
}

extern "C" int rnnwf_commit_params(rnnwf_handle* h) {
    if (!h) return RNNWF_ERR_INVALID;
    if (int rc = train_sync_params_to_host(h)) return rc;
    for (auto& kv : h->params)
        if (!kv.second.set) return h->fail(RNNWF_ERR_STATE, "parameter '%s' was never set", kv.first.c_str());
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = upload_reset(h)) return rc;      // the images travel through one pinned buffer, asynchronously (handle.h: upload)
    std::vector<char> img;
    if (int rc = model_pack_image(h, img)) return rc;
    if (int rc = ensure(h, h->wimg, img.size())) return rc;
    if (int rc = upload(h, h->wimg.p, img.data(), img.size())) return rc;
    h->committed = true;
    h->last_ns = 0;               // resident batch belongs to the old weights
    grad_invalidate(h);
    return RNNWF_OK;
}

// -------------------------------------------------------------------------------------------------
// common helpers
// -------------------------------------------------------------------------------------------------
static int check_ready(rnnwf_handle* h) {
    if (!h) return RNNWF_ERR_INVALID;
    if (!h->committed) return h->fail(RNNWF_ERR_STATE, "parameters not committed (call rnnwf_commit_params)");
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    return 0;
}

int rnnwf::upload_samples(rnnwf_handle* h, const int32_t* samples, int64_t B) {
    if (int rc = ensure(h, h->samples_i32, (size_t)B * h->N * 4)) return rc;
    RNNWF_HIP(h, hipMemcpyAsync(h->samples_i32.p, samples, (size_t)B * h->N * 4, hipMemcpyHostToDevice, h->stream));
    return 0;
}

// packs the (B, N) int32 matrix currently held in h->samples_i32
int rnnwf::pack_device(rnnwf_handle* h, int64_t B, DevBuf& bits, int reverse, const int32_t* col_of_pos_dev) {
    const int N = h->N;
    const int W = (N + 31) / 32;
    if (int rc = ensure(h, bits, (size_t)W * B * 4)) return rc;
    dim3 grid((unsigned)((B + 255) / 256), (unsigned)W);
    pack_bits_kernel<<<grid, 256, 0, h->stream>>>((const int32_t*)h->samples_i32.p, B, N, col_of_pos_dev, reverse,
                                                   (uint32_t*)bits.p);
    RNNWF_HIP(h, hipGetLastError());
    return 0;
}

int rnnwf::upload_and_pack(rnnwf_handle* h, const int32_t* samples, int64_t B, DevBuf& bits, int reverse,
                           const int32_t* col_of_pos_dev) {
    if (int rc = upload_samples(h, samples, B)) return rc;
    return pack_device(h, B, bits, reverse, col_of_pos_dev);
}

// bits -> h->samples_i32 (device), optionally on to the host
int rnnwf::unpack_device(rnnwf_handle* h, const DevBuf& bits, int64_t B, const int32_t* pos_of_col_dev) {
    const int N = h->N;
    if (int rc = ensure(h, h->samples_i32, (size_t)B * N * 4)) return rc;
    const int64_t total = B * N;
    unpack_bits_kernel<<<(unsigned)((total + 255) / 256), 256, 0, h->stream>>>((const uint32_t*)bits.p, B, N,
                                                                                pos_of_col_dev, (int32_t*)h->samples_i32.p);
    RNNWF_HIP(h, hipGetLastError());
    return 0;
}

int rnnwf::unpack_and_download(rnnwf_handle* h, const DevBuf& bits, int64_t B, int32_t* out,
                               const int32_t* pos_of_col_dev) {
    if (int rc = unpack_device(h, bits, B, pos_of_col_dev)) return rc;
    RNNWF_HIP(h, hipMemcpyAsync(out, h->samples_i32.p, (size_t)B * h->N * 4, hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int rnnwf::run_moments(rnnwf_handle* h, const void* eloc_dev, int64_t ns, bool complex_f32, double* moments_host) {
    if (int rc = ensure(h, h->moments, 4 * sizeof(double))) return rc;
    {
        TimedLaunch tl(h, 2);
        // single device: the kernel writes the four moments into pinned host memory itself (read behind the step's one stream
        // sync; saves the copy launch - 4 us of config 1's 70); with the in-step all-reduce they travel device -> RCCL -> copy
        double* direct = (moments_host && !h->reduce_in_step) ? (double*)h->pinned_dev : nullptr;
        if (!moments_host && !h->reduce_in_step && h->moments_direct) direct = h->moments_direct;      // rnnwf_train_steps: iteration k's row of its pinned table
        if (complex_f32)
            moments_kernel<float><<<1, 1024, 0, h->stream>>>((const float*)eloc_dev, ns, 2, 1, (double*)h->moments.p, direct);
        else
            moments_kernel<double><<<1, 1024, 0, h->stream>>>((const double*)eloc_dev, ns, 1, 0, (double*)h->moments.p, direct);
    }
    RNNWF_HIP(h, hipGetLastError());
    if (h->reduce_in_step)
        if (int rc = comm_allreduce_device(h, h->moments.p, 4)) return rc;
    if (moments_host) {
        if (h->reduce_in_step)
            RNNWF_HIP(h, hipMemcpyAsync(h->pinned, h->moments.p, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        RNNWF_HIP(h, hipStreamSynchronize(h->stream));
        memcpy(moments_host, h->pinned, 4 * sizeof(double));
    }
    return 0;
}

int rnnwf::run_tfim_eloc(rnnwf_handle* h, const uint32_t* bits, const double* lpq, int64_t ns, int Nx, int Ny,
                         const int32_t* pos_of_site_dev, const double* Jz_dev, double Bx, double* eloc_dev) {
    TimedLaunch tl(h, 2);
    tfim_eloc_kernel<<<(unsigned)((ns + kElocSamples - 1) / kElocSamples), kElocSamples * kElocGroups, 0, h->stream>>>(
        bits, lpq, ns, Nx, Ny, pos_of_site_dev, Jz_dev, Bx, eloc_dev);
    RNNWF_HIP(h, hipGetLastError());
    return 0;
}

int rnnwf::run_parity_combine(rnnwf_handle* h, const double* a, const double* b, int64_t n, double* out) {
    parity_combine_kernel<<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(a, b, n, out);
    RNNWF_HIP(h, hipGetLastError());
    return 0;
}
int rnnwf::run_parity_share(rnnwf_handle* h, double* lpF, double* lpR, int64_t n) {
    parity_share_kernel<<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(lpF, lpR, n);
    RNNWF_HIP(h, hipGetLastError());
    return 0;
}

int rnnwf::model_pack_image(rnnwf_handle* h, std::vector<char>& img) {
    if (h->model == RNNWF_MODEL_MDRNN2D) return mdrnn_pack_image(h, img);
    if (h->model == RNNWF_MODEL_CRNN_U1) return crnn_pack_image(h, img);
    return prnn_pack_image(h, img);
}

// -------------------------------------------------------------------------------------------------
// public compute entry points: dispatch on the model
// -------------------------------------------------------------------------------------------------
static bool is_prnn(const rnnwf_handle* h) {
    return h->model == RNNWF_MODEL_GRU1D || h->model == RNNWF_MODEL_GRU1D_PARITY || h->model == RNNWF_MODEL_GRU1D_F64;
}

extern "C" int rnnwf_sample(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset,
                            int32_t* out_samples, double* out_log) {
    if (int rc = check_ready(h)) return rc;
    if (ns < 1 || !out_samples) return h->fail(RNNWF_ERR_INVALID, "rnnwf_sample: numsamples must be >= 1 and out_samples non-null");
    if (is_prnn(h)) return prnn_sample(h, ns, seed, step, offset, out_samples, out_log);
    if (h->model == RNNWF_MODEL_CRNN_U1) return crnn_sample(h, ns, seed, step, offset, out_samples, out_log);
    return mdrnn_sample(h, ns, seed, step, offset, out_samples, out_log);
}

extern "C" int rnnwf_log_prob(rnnwf_handle* h, const int32_t* samples, int64_t B, double* out) {
    if (int rc = check_ready(h)) return rc;
    if (B < 0 || (B > 0 && (!samples || !out))) return h->fail(RNNWF_ERR_INVALID, "rnnwf_log_prob: bad arguments");
    if (B == 0) return RNNWF_OK;
    if (is_prnn(h)) return prnn_log_prob(h, samples, B, out);
    if (h->model == RNNWF_MODEL_CRNN_U1) return crnn_log_amp(h, samples, B, nullptr, out);
    return mdrnn_log_prob(h, samples, B, out);
}

extern "C" int rnnwf_log_amp(rnnwf_handle* h, const int32_t* samples, int64_t B, float* out_re_im) {
    if (int rc = check_ready(h)) return rc;
    if (h->model != RNNWF_MODEL_CRNN_U1) return h->fail(RNNWF_ERR_INVALID, "rnnwf_log_amp: only the complex RNN has amplitudes");
    if (B < 0 || (B > 0 && (!samples || !out_re_im))) return h->fail(RNNWF_ERR_INVALID, "rnnwf_log_amp: bad arguments");
    if (B == 0) return RNNWF_OK;
    return crnn_log_amp(h, samples, B, out_re_im, nullptr);
}

extern "C" int rnnwf_tfim_eloc(rnnwf_handle* h, const int32_t* samples, int64_t ns, const double* Jz, double Bx,
                               double* eloc, double* log_probs) {
    if (int rc = check_ready(h)) return rc;
    if (h->model != RNNWF_MODEL_GRU1D && h->model != RNNWF_MODEL_GRU1D_PARITY)
        return h->fail(RNNWF_ERR_INVALID, "rnnwf_tfim_eloc: needs a 1D positive RNN handle");
    if (ns < 1 || !samples || !Jz || !eloc) return h->fail(RNNWF_ERR_INVALID, "rnnwf_tfim_eloc: bad arguments");
    return prnn_tfim_eloc(h, samples, ns, 1, h->N, Jz, Bx, eloc, log_probs);
}

extern "C" int rnnwf_tfim2d_eloc(rnnwf_handle* h, const int32_t* samples, int64_t ns, const double* Jz, double Bx,
                                 double* eloc, double* log_probs) {
    if (int rc = check_ready(h)) return rc;
    if (ns < 1 || !samples || !Jz || !eloc) return h->fail(RNNWF_ERR_INVALID, "rnnwf_tfim2d_eloc: bad arguments");
    if (h->model == RNNWF_MODEL_GRU1D_F64) return prnn_tfim_eloc(h, samples, ns, h->Nx, h->Ny, Jz, Bx, eloc, log_probs);
    if (h->model == RNNWF_MODEL_MDRNN2D) return mdrnn_tfim_eloc(h, samples, ns, Jz, Bx, eloc, log_probs);
    return h->fail(RNNWF_ERR_INVALID, "rnnwf_tfim2d_eloc: needs a 2D handle (GRU1D_F64 or MDRNN2D)");
}

extern "C" int rnnwf_j1j2_eloc(rnnwf_handle* h, const int32_t* samples, int64_t ns, const double* J1, const double* J2,
                               const double* Bz, int32_t periodic, int32_t marshall, float* eloc, int64_t* ncon) {
    if (int rc = check_ready(h)) return rc;
    if (h->model != RNNWF_MODEL_CRNN_U1) return h->fail(RNNWF_ERR_INVALID, "rnnwf_j1j2_eloc: needs a complex RNN handle");
    if (ns < 1 || !samples || !J1 || !J2 || !Bz || !eloc) return h->fail(RNNWF_ERR_INVALID, "rnnwf_j1j2_eloc: bad arguments");
    return crnn_j1j2_eloc(h, samples, ns, J1, J2, Bz, periodic, marshall, eloc, ncon);
}

extern "C" int rnnwf_vmc_step(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset,
                              const double* couplings, int64_t n_couplings, int32_t* out_samples, void* out_eloc,
                              double* moments) {
    if (int rc = check_ready(h)) return rc;
    if (ns < 1 || !couplings || !moments) return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_step: bad arguments");
    if (is_prnn(h)) {
        if (n_couplings != h->N + 1) return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_step: TFIM needs N+1 couplings (Jz, Bx)");
        return prnn_vmc_step(h, ns, seed, step, offset, couplings, out_samples, (double*)out_eloc, moments);
    }
    if (h->model == RNNWF_MODEL_MDRNN2D) {
        if (n_couplings != h->N + 1) return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_step: TFIM needs Nx*Ny+1 couplings (Jz, Bx)");
        return mdrnn_vmc_step(h, ns, seed, step, offset, couplings, out_samples, (double*)out_eloc, moments);
    }
    if (n_couplings != 3 * h->N + 2) return h->fail(RNNWF_ERR_INVALID, "rnnwf_vmc_step: J1J2 needs 3N+2 couplings");
    return crnn_vmc_step(h, ns, seed, step, offset, couplings, out_samples, (float*)out_eloc, moments);
}

extern "C" int rnnwf_load_batch(rnnwf_handle* h, const int32_t* samples, int64_t ns, const void* eloc) {
    if (int rc = check_ready(h)) return rc;
    if (ns < 1 || !samples || !eloc) return h->fail(RNNWF_ERR_INVALID, "rnnwf_load_batch: bad arguments");
    h->last_ns = 0;
    int rc;
    if (is_prnn(h)) rc = prnn_load_batch(h, samples, ns);
    else if (h->model == RNNWF_MODEL_CRNN_U1) rc = crnn_load_batch(h, samples, ns);
    else rc = mdrnn_load_batch(h, samples, ns);
    if (rc) return rc;
    const size_t bytes = (size_t)ns * 8;                  // float64 per sample, or complex64 = two float32 per sample
    RNNWF_HIP(h, hipMemcpyAsync(h->eloc.p, eloc, bytes, hipMemcpyHostToDevice, h->stream));
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    h->last_ns = ns;
    h->last_has_ckpt = true;
    return RNNWF_OK;
}

// -------------------------------------------------------------------------------------------------
// measurement
// -------------------------------------------------------------------------------------------------
extern "C" int rnnwf_timing_enable(rnnwf_handle* h, int32_t on) {
    if (!h) return RNNWF_ERR_INVALID;
    h->timing_on = on != 0;
    h->timing_mask = on == 2 ? 2 : 31;      // 2: events around the dominant (flip / swap) pass only - two per step instead of ten
    return RNNWF_OK;
}

static int drain_timers(rnnwf_handle* h) {
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    for (auto& t : h->timers) {
        for (auto& ev : t.pending) {
            float ms = 0.f;
            RNNWF_HIP(h, hipEventElapsedTime(&ms, ev.first, ev.second));
            t.total_ms += ms;
            t.launches += 1;
            t.pool.push_back(ev);
        }
        t.pending.clear();
    }
    return 0;
}

extern "C" int rnnwf_timing_reset(rnnwf_handle* h) {
    if (!h) return RNNWF_ERR_INVALID;
    if (int rc = drain_timers(h)) return rc;
    for (auto& t : h->timers) { t.total_ms = 0.0; t.launches = 0; }
    h->work[0] = h->work[1] = 0.0;
    return RNNWF_OK;
}

extern "C" int rnnwf_timing_get(rnnwf_handle* h, int32_t id, double* total_ms, int64_t* launches, double* work) {
    if (!h || id < 0 || id > 4) return RNNWF_ERR_INVALID;
    if (int rc = drain_timers(h)) return rc;
    if (total_ms) *total_ms = h->timers[id].total_ms;
    if (launches) *launches = h->timers[id].launches;
    if (work) { work[0] = h->work[0]; work[1] = h->work[1]; }
    return RNNWF_OK;
}

extern "C" const char* rnnwf_engine_name(const rnnwf_handle* h) {
    if (!h) return "";
    if (h->f64) return "f64mfma";
    if (h->last_flip_engine >= 0 && h->model != RNNWF_MODEL_CRNN_U1) return h->last_flip_engine ? "bf16x3" : "f32mfma";
    return h->engine_split ? "bf16x3" : "f32mfma";
}

extern "C" int rnnwf_synchronize(rnnwf_handle* h) {
    if (!h) return RNNWF_ERR_INVALID;
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    RNNWF_HIP(h, hipDeviceSynchronize());
    return RNNWF_OK;
}

extern "C" int rnnwf_device_info(rnnwf_handle* h, int32_t* cu, int32_t* mhz, int64_t* hbm, char* name64) {
    if (!h) return RNNWF_ERR_INVALID;
    hipDeviceProp_t prop;
    RNNWF_HIP(h, hipGetDeviceProperties(&prop, h->cfg.device));
    if (cu) *cu = prop.multiProcessorCount;
    if (mhz) *mhz = prop.clockRate / 1000;
    if (hbm) *hbm = (int64_t)prop.totalGlobalMem;
    if (name64) { strncpy(name64, prop.name, 63); name64[63] = 0; }
    return RNNWF_OK;
}
