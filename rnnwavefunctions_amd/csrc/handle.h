// handle.h - the opaque handle behind include/rnnwf.h (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rnnwf.h"

namespace rnnwf {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double total_ms = 0.0;
    int64_t launches = 0;
};

// Environment switches, read ONCE in rnnwf_create (never on the per-step path).
struct Knobs {
    int engine = 0;               // RNNWF_ENGINE: 0 default, 1 "f32" (f32-input MFMA everywhere), 2 "bf16x3" (pinned), 3 "bf16x3-serial" (pinned, 4-wave kernel without the ping-pong: A/B), 4 "bf16x3-hipcc" (pinned; 69..100 units: the compiler-scheduled riders step instead of the generated asm block: A/B), 5 "bf16x3-asm32" (pinned; 69..100 units: the 32x32x16 asm step instead of the 16x16x32 one: A/B), 7 "bf16x3-n16" (pinned; 37..52 units, positive RNN: the 16x16x32 riders form instead of the 32x32x16 ping-pong kernel: A/B, 10 % slower)
    bool no_coop = false;         // RNNWF_NO_COOP=1: base pass always on the one-wave-per-block kernel
    bool base_f32 = false;        // RNNWF_BASE=f32: the base pass keeps the f32-input MFMA (no bf16 cooperative kernel): A/B, and the
                                  // bit-identity test of the two f32 kernels
    bool md_prefetch = false;     // RNNWF_MDRNN_PREFETCH=1: the MDRNN flip pass with the LDS-DMA prefetch of the vertical state (measured 1.7 % slower at config 4; A/B only)
    size_t state_budget = 0;      // RNNWF_STATE_BUDGET_MB << 20 (0: the family's default)
    int ablate = 0, ablate_base = 0;   // RNNWF_ABLATE / RNNWF_ABLATE_BASE: only in -DRNNWF_DIAGNOSTICS builds (tools/)
};

// Device-resident training (train.hip): the parameters, Adam's moments and the flat gradient live on the device; after every update
// the kernels' weight images are rebuilt there by replaying the host packers' recorded element tables (pack_value.h).
struct TrainImage {
    DevBuf table;                 // PackEntry[n]
    int64_t n = 0;
    DevBuf* target = nullptr;     // the image it rebuilds (h->wimg, h->wsplit, ...)
};
struct TrainState {
    bool built = false, supported = false;
    std::string why;              // why not supported
    DevBuf P, M, V, G;            // flat f64: parameters (order of rnnwf_set_params_flat), Adam's m and v, gradient
    DevBuf gidx;                  // int32[nparams]: +-(k + 1) = element k of the dW image (sign: the head's logit difference), 0: none
    int64_t nparams = 0;
    size_t dw_count = 0;
    bool dw_f64 = false;
    TrainImage img[8];
    int nimg = 0;
    int64_t adam_t = 0;           // updates applied so far (tf.train.AdamOptimizer's beta powers are beta^(t+1))
    bool host_newer = true;       // the host copy of the parameters changed since the device copy was made
    bool dev_newer = false;       // the device copy was updated by an optimizer step the host copy has not seen
    void* mom_host = nullptr;     // pinned [kMaxSteps][4] doubles: the moments of the K steps of one rnnwf_train_steps call
    void* mom_host_dev = nullptr; // the same memory as the device addresses it
    DevBuf combo;                 // the tables of all images back to back: ONE re-pack launch per update (train.hip: repack_all_kernel)
    int combo_nimg = 0;           // images the combined table was built from
};

struct ParamSpec {
    std::vector<int64_t> shape; // as the caller sees it (the reference's TF variable)
    std::vector<double> value;  // stored in f64, converted to the model type when packed.  Layers of unequal width are held PADDED
                                // to the widest layer (zeros: a padded unit stays exactly 0 and feeds nothing), so that the packers
                                // and kernels see one width; `slot` maps the caller's flat index to the padded one.
    std::vector<int64_t> slot;  // size = number of elements the caller sees
    bool set = false;
};

}  // namespace rnnwf

struct rnnwf_handle {
    rnnwf_config cfg{};
    int model = 0;
    bool f64 = false;
    int H = 0;       // num_units
    int NFULL = 0;   // padded hidden size = 16 NFULL + 4
    int NL = 1;      // stacked GRU layers (len(units)); > 1 only for the f32 1D positive RNN
    int N = 0;       // chain length (nx * ny)
    int Nx = 1, Ny = 1;
    int cu_count = 0;
    std::string err;
    hipStream_t stream = nullptr;
    std::map<std::string, rnnwf::ParamSpec> params;
    std::map<std::string, std::vector<int32_t>> param_flat;   // per tensor: padded index -> index in the flat parameter vector
                                                              // (order of rnnwf_set_params_flat), -1 for padding (pack_value.h)
    bool committed = false;

    // device buffers (grown on demand, owned by the handle)
    rnnwf::DevBuf wimg, samples_i32, bits, bits2, hck, lpq, lpq2, out_lp, out_lp2, eloc, moments, coupl, maps;
    // J1J2 / cRNN
    rnnwf::DevBuf camp, tiles, tile_count, cbase, cout;
    // MDRNN
    rnnwf::DevBuf rowbuf;
    // gradient (grad.hip)
    rnnwf::DevBuf wbwd, gradP, gradQ, gradW;
    bool wbwd_valid = false;      // the backward image matches the committed parameters (re-packed by the next gradient call otherwise)
    rnnwf::DevBuf gradPart, gradHeadPart;    // per-block partial sums of the weight-gradient GEMM; per-wave head-row sums
    std::vector<double> coupl_host;   // what h->coupl holds (couplings rarely change between steps: skip the upload)
    rnnwf::DevBuf gradDX[2];      // stacked layers: dL/dx of one layer's pass = dL/dh input of the pass below
    // bf16x3 engine (split_core.h): second weight image; engine_split = use it for the flip pass
    rnnwf::DevBuf wsplit;
    rnnwf::DevBuf wsplit16;       // image of the 16x16x32 form of the flip pass at 69..100 units (split16_core.h)
    rnnwf::DevBuf wbasebf;        // bf16x3 A fragments of the cooperative base pass (layout.h: BaseBfLayout); valid iff base_bf
    rnnwf::DevBuf wsplit_up[RNNWF_MAX_LAYERS - 1];   // stacked layers on the bf16x3 engine: image of layer l in [l - 1] (split_core.h: SplitUpperLayout)
    rnnwf::DevBuf xrec[2];        // their layer pipeline: per-step state records of one layer, read by the kernel of the layer above
    rnnwf::TrainState train;      // device-resident training (train.hip)
    bool base_bf = false;
    bool engine_split = false;
    bool engine_forced = false;   // RNNWF_ENGINE=bf16x3: no small-batch fallback to the f32-input MFMA
    int64_t call_ns = 0;          // samples of the whole API call (a call may run several passes: one engine for all)
    int last_flip_engine = -1;    // engine of the last flip / swap launch (1 bf16x3, 0 f32-input MFMA), -1 none yet
    std::map<std::string, std::vector<double>> grads;
    int64_t last_ns = 0;          // batch of the last rnnwf_vmc_step still resident (bits, hck, eloc)
    bool last_has_ckpt = false;
    void* pinned = nullptr;  // small pinned staging (moments)
    void* pinned_dev = nullptr;   // the same memory as the device addresses it: kernels write the step's 32 + 24 result bytes there directly
    double* moments_direct = nullptr;   // device address of the pinned row the moments kernel also writes to (set per iteration by rnnwf_train_steps)
    bool j1j2_cnt_clean = false;  // the J1-J2 item counters are zero (left so by the last assembly kernel)
    void* staging = nullptr; // pinned staging of the host-side all-reduces (comm.hip) and of the gradient's download, grown on demand
    size_t staging_cap = 0;
    void* upbuf = nullptr;   // pinned buffer the weight images travel through on their way to the device (upload(), below)
    size_t upbuf_cap = 0, upbuf_off = 0;
    rnnwf::DevBuf reduce_scratch;

    bool timing_on = false;
    int timing_mask = 31;    // which kernel ids get HIP events (rnnwf_timing_enable: 1 = all, 2 = the dominant pass only)
    rnnwf::KernelTimer timers[5];   // 0 base pass, 1 flip / swap pass, 2 assembly, 3 gradient back-propagation, 4 weight-gradient GEMM
    double work[2] = {0.0, 0.0};

    rnnwf::Knobs knobs;
    // resident blocks per CU of each kernel this handle has launched (the dynamic-LDS attribute is per device and the
    // handle is single-threaded by contract, so the cache lives here: handles on other host threads share nothing)
    std::map<const void*, int> occupancy;

    void* comm = nullptr;  // ncclComm_t
    int rank = 0, nranks = 1;
    bool reduce_in_step = false;   // rnnwf_vmc_step returns the moments summed over all ranks (one RCCL all-reduce on the
                                   // stream, before the single host synchronisation of the step)

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define RNNWF_HIP(h, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return (h)->fail(RNNWF_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

namespace rnnwf {

inline int ensure(rnnwf_handle* h, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) RNNWF_HIP(h, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    const size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) return h->fail(RNNWF_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    b.cap = want;
    return 0;
}

// Pinned staging of `bytes` (grown on demand, owned by the handle): a pageable source or destination makes hipMemcpyAsync a hidden
// synchronous copy through the runtime's own bounce buffer.
inline int ensure_staging(rnnwf_handle* h, size_t bytes) {
    if (bytes <= h->staging_cap) return 0;
    if (h->staging) RNNWF_HIP(h, hipHostFree(h->staging));
    h->staging = nullptr;
    h->staging_cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&h->staging, want, hipHostMallocDefault);
    if (e != hipSuccess) return h->fail(RNNWF_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    h->staging_cap = want;
    return 0;
}

// Host -> device copy of a freshly packed image, asynchronous on the handle's stream: the bytes are copied into the handle's
// pinned upload buffer first (bump-allocated; upload_reset() at the start of a commit waits for the previous commit's copies,
// which have long landed), so the caller's vector may die at once and nothing blocks.  A training iteration re-packs three to
// four images; with blocking hipMemcpy each cost 10-20 us of host time (tools/iter_breakdown.py).
inline int upload_reset(rnnwf_handle* h) {
    if (h->upbuf_off) RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    h->upbuf_off = 0;
    return 0;
}
inline int upload(rnnwf_handle* h, void* dst, const void* src, size_t bytes) {
    const size_t off = (h->upbuf_off + 255) & ~(size_t)255;
    if (off + bytes > h->upbuf_cap) {                     // full: wait for what is in flight, then start over - in the same buffer when the
        RNNWF_HIP(h, hipStreamSynchronize(h->stream));    // image fits it (a caller that packs again and again without a commit in between, which
        if (bytes <= h->upbuf_cap) {                      // alone resets the offset, must not double the buffer at every overflow), else a larger one
            h->upbuf_off = 0;
            return upload(h, dst, src, bytes);
        }
        if (h->upbuf) RNNWF_HIP(h, hipHostFree(h->upbuf));
        h->upbuf = nullptr;
        h->upbuf_cap = 0;
        const size_t want = std::max<size_t>((size_t)1 << 20, 2 * (off + bytes));
        hipError_t e = hipHostMalloc(&h->upbuf, want, hipHostMallocDefault);
        if (e != hipSuccess) return h->fail(RNNWF_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        h->upbuf_cap = want;
        h->upbuf_off = 0;
        return upload(h, dst, src, bytes);
    }
    memcpy((char*)h->upbuf + off, src, bytes);
    RNNWF_HIP(h, hipMemcpyAsync(dst, (char*)h->upbuf + off, bytes, hipMemcpyHostToDevice, h->stream));
    h->upbuf_off = off + bytes;
    return 0;
}

// Resident blocks per CU of kernel `fn` at `threads` per block and `lds` dynamic bytes; raises the kernel's
// dynamic-LDS limit on first use.  Cached per handle.
inline int blocks_per_cu(rnnwf_handle* h, const void* fn, int threads, size_t lds, int* out) {
    auto it = h->occupancy.find(fn);
    if (it != h->occupancy.end()) { *out = it->second; return 0; }
    RNNWF_HIP(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nb = 0;
    RNNWF_HIP(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds));
    nb = nb < 1 ? 1 : nb;
    h->occupancy[fn] = nb;
    *out = nb;
    return 0;
}

// HIP-event bracket around one launch on the handle's stream
struct TimedLaunch {
    rnnwf_handle* h;
    int id;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    bool on() const { return h->timing_on && ((h->timing_mask >> id) & 1); }
    TimedLaunch(rnnwf_handle* h_, int id_) : h(h_), id(id_) {
        if (!on()) return;
        KernelTimer& t = h->timers[id];
        if (!t.pool.empty()) {
            ev = t.pool.back();
            t.pool.pop_back();
        } else {
            hipEventCreate(&ev.first);
            hipEventCreate(&ev.second);
        }
        hipEventRecord(ev.first, h->stream);
    }
    ~TimedLaunch() {
        if (!on()) return;
        hipEventRecord(ev.second, h->stream);
        h->timers[id].pending.push_back(ev);
    }
};

}  // namespace rnnwf
