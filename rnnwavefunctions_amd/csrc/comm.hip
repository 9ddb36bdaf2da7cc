// comm.hip - the one collective of the path: an RCCL all-reduce of the energy moments.
// librccl.so is loaded lazily with dlopen so that single-GPU use never pays for it.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "models.h"

namespace {
typedef struct { char internal[128]; } ncclUniqueId_;
typedef int (*GetUniqueId_t)(ncclUniqueId_*);
typedef int (*CommInitRank_t)(void**, int, ncclUniqueId_, int);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroy_t)(void*);
typedef const char* (*GetErrorString_t)(int);
typedef int (*CommCount_t)(void*, int*);
typedef int (*CommUserRank_t)(void*, int*);

struct Rccl {
    void* lib = nullptr;
    GetUniqueId_t get_unique_id = nullptr;
    CommInitRank_t comm_init_rank = nullptr;
    AllReduce_t all_reduce = nullptr;
    CommDestroy_t comm_destroy = nullptr;
    GetErrorString_t error_string = nullptr;
    CommCount_t comm_count = nullptr;
    CommUserRank_t comm_user_rank = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        // The RCCL of the ROCm installation this library's HIP runtime comes from, by absolute path first: a process
        // that imported PyTorch already holds torch's bundled librccl (built against torch's own bundled HIP runtime),
        // and a by-name dlopen would hand that one back - two HIP runtimes would then share device pointers.
        std::string rocm = getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "/opt/rocm";
        const std::string abs1 = rocm + "/lib/librccl.so.1";
        const char* names[] = {abs1.c_str(), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
        for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) { err = std::string("dlopen(librccl.so) failed: ") + dlerror(); return false; }
        get_unique_id = (GetUniqueId_t)dlsym(lib, "ncclGetUniqueId");
        comm_init_rank = (CommInitRank_t)dlsym(lib, "ncclCommInitRank");
        all_reduce = (AllReduce_t)dlsym(lib, "ncclAllReduce");
        comm_destroy = (CommDestroy_t)dlsym(lib, "ncclCommDestroy");
        error_string = (GetErrorString_t)dlsym(lib, "ncclGetErrorString");
        comm_count = (CommCount_t)dlsym(lib, "ncclCommCount");
        comm_user_rank = (CommUserRank_t)dlsym(lib, "ncclCommUserRank");
        if (!get_unique_id || !comm_init_rank || !all_reduce || !comm_destroy) { err = "librccl.so lacks the nccl* symbols"; return false; }
        return true;
    }
} g_rccl;

constexpr int kNcclFloat64 = 8;  // ncclDouble
constexpr int kNcclSum = 0;
static_assert(sizeof(ncclUniqueId_) == RNNWF_UNIQUE_ID_BYTES, "unique id size");
}  // namespace

extern "C" int rnnwf_comm_unique_id(void* id_out) {
    if (!id_out || !g_rccl.load()) return RNNWF_ERR_COMM;
    ncclUniqueId_ id;
    if (g_rccl.get_unique_id(&id) != 0) return RNNWF_ERR_COMM;
    memcpy(id_out, &id, sizeof id);
    return RNNWF_OK;
}

extern "C" int rnnwf_comm_init(rnnwf_handle* h, const void* id, int32_t rank, int32_t nranks) {
    if (!h || !id) return RNNWF_ERR_INVALID;
    if (rank < 0 || rank >= nranks) return h->fail(RNNWF_ERR_INVALID, "rnnwf_comm_init: rank %d outside [0,%d)", rank, nranks);
    if (!g_rccl.load()) return h->fail(RNNWF_ERR_COMM, "%s", g_rccl.err.c_str());
    if (h->comm) rnnwf_comm_destroy(h);
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    ncclUniqueId_ uid;
    memcpy(&uid, id, sizeof uid);
    const int rc = g_rccl.comm_init_rank(&h->comm, nranks, uid, rank);
    if (rc != 0) {
        h->comm = nullptr;
        return h->fail(RNNWF_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.error_string ? g_rccl.error_string(rc) : "?");
    }
    h->rank = rank;
    h->nranks = nranks;
    return RNNWF_OK;
}

// What the communicator itself says (ncclCommCount / ncclCommUserRank), not what the caller passed to rnnwf_comm_init:
// bench.py prints it so that N independent 1-rank runs cannot pass for one N-rank run.
extern "C" int rnnwf_comm_info(rnnwf_handle* h, int32_t* nranks, int32_t* rank, int32_t* device) {
    if (!h) return RNNWF_ERR_INVALID;
    int n = 1, r = 0;
    if (h->comm) {
        if (!g_rccl.comm_count || !g_rccl.comm_user_rank) return h->fail(RNNWF_ERR_COMM, "librccl.so lacks ncclCommCount / ncclCommUserRank");
        int rc = g_rccl.comm_count(h->comm, &n);
        if (rc == 0) rc = g_rccl.comm_user_rank(h->comm, &r);
        if (rc != 0) return h->fail(RNNWF_ERR_COMM, "ncclCommCount failed: %s", g_rccl.error_string ? g_rccl.error_string(rc) : "?");
    }
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (device) *device = h->cfg.device;
    return RNNWF_OK;
}

// in-stream sum over the ranks of `count` doubles already on the device (no host round trip, no synchronisation)
int rnnwf::comm_allreduce_device(rnnwf_handle* h, void* dev, size_t count) {
    if (!h->comm) return 0;
    const int rc = g_rccl.all_reduce(dev, dev, count, kNcclFloat64, kNcclSum, h->comm, h->stream);
    if (rc != 0) return h->fail(RNNWF_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.error_string ? g_rccl.error_string(rc) : "?");
    return 0;
}

extern "C" int rnnwf_comm_reduce_in_step(rnnwf_handle* h, int32_t on) {
    if (!h) return RNNWF_ERR_INVALID;
    if (on && !h->comm) return h->fail(RNNWF_ERR_STATE, "rnnwf_comm_reduce_in_step: communicator not initialised");
    h->reduce_in_step = on != 0;
    return RNNWF_OK;
}

// `count` doubles in pinned staging -> device scratch -> ncclAllReduce(sum) on the handle's stream -> staging; one host sync
static int allreduce_staged(rnnwf_handle* h, size_t count) {
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = rnnwf::ensure(h, h->reduce_scratch, count * sizeof(double))) return rc;
    RNNWF_HIP(h, hipMemcpyAsync(h->reduce_scratch.p, h->staging, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const int rc = g_rccl.all_reduce(h->reduce_scratch.p, h->reduce_scratch.p, count, kNcclFloat64, kNcclSum, h->comm, h->stream);
    if (rc != 0) return h->fail(RNNWF_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.error_string ? g_rccl.error_string(rc) : "?");
    RNNWF_HIP(h, hipMemcpyAsync(h->staging, h->reduce_scratch.p, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RNNWF_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int rnnwf_allreduce_f64(rnnwf_handle* h, double* data, int64_t count) {
    if (!h || count < 0 || (count > 0 && !data)) return RNNWF_ERR_INVALID;
    if (!h->comm) {
        if (h->nranks == 1) return RNNWF_OK;  // single device: the local sums are the global ones
        return h->fail(RNNWF_ERR_STATE, "rnnwf_allreduce_f64: communicator not initialised");
    }
    if (count == 0) return RNNWF_OK;
    if (int rc = rnnwf::ensure_staging(h, (size_t)count * sizeof(double))) return rc;
    memcpy(h->staging, data, (size_t)count * sizeof(double));
    if (int rc = allreduce_staged(h, (size_t)count)) return rc;
    memcpy(data, h->staging, (size_t)count * sizeof(double));
    return RNNWF_OK;
}

extern "C" int rnnwf_allreduce_moments(rnnwf_handle* h, double* moments, int32_t count) {
    if (!h || !moments || count < 1 || count > 64) return RNNWF_ERR_INVALID;
    return rnnwf_allreduce_f64(h, moments, count);
}

extern "C" int rnnwf_allreduce_grads(rnnwf_handle* h) {
    if (!h) return RNNWF_ERR_INVALID;
    if (h->grads.empty()) return h->fail(RNNWF_ERR_STATE, "rnnwf_allreduce_grads: no gradients (call rnnwf_vmc_gradient first)");
    if (!h->comm) {
        if (h->nranks == 1) return RNNWF_OK;
        return h->fail(RNNWF_ERR_STATE, "rnnwf_allreduce_grads: communicator not initialised");
    }
    RNNWF_HIP(h, hipSetDevice(h->cfg.device));
    {   // the gradient is flattened and summed on the device, in-stream (train.hip); the staged road below is the fall-back
        const int done = rnnwf::train_allreduce_grads_device(h);
        if (done < 0) return done;
        if (done == 1) return RNNWF_OK;
    }
    size_t total = 0;
    for (auto& kv : h->grads) total += kv.second.size();      // std::map: same order on every rank
    if (int rc = rnnwf::ensure_staging(h, total * sizeof(double))) return rc;
    double* flat = static_cast<double*>(h->staging);
    size_t off = 0;
    for (auto& kv : h->grads) { std::copy(kv.second.begin(), kv.second.end(), flat + off); off += kv.second.size(); }
    if (int rc = allreduce_staged(h, total)) return rc;
    off = 0;
    for (auto& kv : h->grads) { std::copy(flat + off, flat + off + kv.second.size(), kv.second.begin()); off += kv.second.size(); }
    return RNNWF_OK;
}

extern "C" int rnnwf_comm_destroy(rnnwf_handle* h) {
    if (!h) return RNNWF_ERR_INVALID;
    if (h->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(h->comm);
    h->comm = nullptr;
    h->nranks = 1;
    h->rank = 0;
    h->reduce_in_step = false;
    return RNNWF_OK;
}
