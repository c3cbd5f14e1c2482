// dist.hip -- the multi-GPU side of the boundary for a C caller: one process per GPU, ONE RCCL
// communicator, ONE ncclBroadcast of each packed weight block from the root at start-up over xGMI, and nothing on the
// data path (utterances are independent: batch_norm.c:178-181 uses stored statistics, every sequence owns its state).
//
// RCCL is loaded with dlopen at the first nntk_dist_* call: the library has no link-time dependency on it, and a
// single-GPU caller never touches it.
#include "nntk_common.hpp"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
} R;
std::mutex g_mutex;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;
float *g_stage = nullptr;          // device staging for host blocks
size_t g_stage_n = 0;

int load_rccl() {
    if (R.lib) return 0;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (R.lib) break;
    }
    if (!R.lib) return nntk_fail_msg("nntk_dist: cannot load librccl.so (RCCL is needed only for multi-GPU weight broadcast)");
#define NNTK_RCCL_SYM(field, sym) \
    *(void **)(&R.field) = dlsym(R.lib, sym); \
    if (!R.field) { R.lib = nullptr; return nntk_fail_msg("nntk_dist: librccl.so lacks " sym); }
    NNTK_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    NNTK_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    NNTK_RCCL_SYM(Broadcast, "ncclBroadcast")
    NNTK_RCCL_SYM(AllReduce, "ncclAllReduce")
    NNTK_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    NNTK_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef NNTK_RCCL_SYM
    return 0;
}
int rccl_fail(const char *what, ncclResult_t rc) {
    char msg[256];
    snprintf(msg, sizeof(msg), "RCCL error in %s: %s", what, R.GetErrorString ? R.GetErrorString(rc) : "?");
    return nntk_fail_msg(msg);
}
#define NNTK_RCCL_TRY(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return rccl_fail(#expr, _r); } while (0)
}  // namespace

extern "C" {

// rank 0 creates the 128-byte id; the caller carries it to the other ranks (file, environment, socket, MPI ...)
int nntk_shim_dist_unique_id(unsigned char *id128) {
    std::lock_guard<std::mutex> lk(g_mutex);
    if (load_rccl()) return -1;
    ncclUniqueId id;
    NNTK_RCCL_TRY(R.GetUniqueId(&id));
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

// collective: every rank calls it, after selecting its GPU (nntk_hip_set_device)
int nntk_shim_dist_init(const unsigned char *id128, int rank, int world) {
    std::lock_guard<std::mutex> lk(g_mutex);
    if (world < 1 || rank < 0 || rank >= world) return nntk_fail_msg("nntk_dist_init: need 0 <= rank < world_size");
    if (g_comm) return nntk_fail_msg("nntk_dist_init: already initialised (nntk_dist_finalize first)");
    if (load_rccl()) return -1;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    NNTK_RCCL_TRY(R.CommInitRank(&g_comm, world, id, rank));
    g_rank = rank;
    g_world = world;
    return 0;
}
int nntk_shim_dist_rank(void) { return g_rank; }
int nntk_shim_dist_world(void) { return g_world; }

// In-place broadcast of a HOST block (a layer's caller-visible weight block) from `root`: staged through device memory
// so that the bytes travel GPU to GPU over xGMI.  Blocking.  Without a communicator it is a no-op.
int nntk_shim_dist_broadcast_host(float *block, size_t n, int root) {
    std::lock_guard<std::mutex> lk(g_mutex);
    if (!g_comm || n == 0) return 0;       // no communicator = single-GPU caller: nothing to do
    if (root < 0 || root >= g_world) return nntk_fail_msg("nntk_dist_broadcast: bad root");
    if (g_stage_n < n) {
        if (g_stage) (void)hipFree(g_stage);
        g_stage = nullptr; g_stage_n = 0;
        NNTK_HIP_TRY(hipMalloc((void **)&g_stage, n * sizeof(float)));
        g_stage_n = n;
    }
    hipStream_t st = nntk_stream();
    if (g_rank == root) NNTK_HIP_TRY(hipMemcpyAsync(g_stage, block, n * sizeof(float), hipMemcpyHostToDevice, st));
    NNTK_RCCL_TRY(R.Broadcast(g_stage, g_stage, n, ncclFloat, root, g_comm, st));
    if (g_rank != root) NNTK_HIP_TRY(hipMemcpyAsync(block, g_stage, n * sizeof(float), hipMemcpyDeviceToHost, st));
    NNTK_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// Data-parallel training ("where gradient all-reduce over xGMI would first appear"): in-place SUM of a gradient
// block over the ranks.  Device form: asynchronous on the calling thread's stream -- a caller overlaps one layer's all-reduce
// with the next layer's backward pass by issuing them from two threads / streams (nntk_hip_set_stream); the device-pointer
// gradient calls (<Layer>CalculateGradientDevice) leave their blocks in HBM for exactly this (note that those calls themselves
// synchronise their stream once per call for the weight upload -- nntoolkitcore_hip.h -- so the overlap is between two host threads
// on two streams, not inside one stream).  Host form: staged and blocking.
// Without a communicator both are no-ops (world size 1: the sum over one rank).
int nntk_shim_dist_allreduce_device(float *d_block, size_t n) {
    std::lock_guard<std::mutex> lk(g_mutex);
    if (!g_comm || n == 0) return 0;
    NNTK_RCCL_TRY(R.AllReduce(d_block, d_block, n, ncclFloat, ncclSum, g_comm, nntk_stream()));
    return 0;
}
int nntk_shim_dist_allreduce_host(float *block, size_t n) {
    std::lock_guard<std::mutex> lk(g_mutex);
    if (!g_comm || n == 0) return 0;
    if (g_stage_n < n) {
        if (g_stage) (void)hipFree(g_stage);
        g_stage = nullptr; g_stage_n = 0;
        NNTK_HIP_TRY(hipMalloc((void **)&g_stage, n * sizeof(float)));
        g_stage_n = n;
    }
    hipStream_t st = nntk_stream();
    NNTK_HIP_TRY(hipMemcpyAsync(g_stage, block, n * sizeof(float), hipMemcpyHostToDevice, st));
    NNTK_RCCL_TRY(R.AllReduce(g_stage, g_stage, n, ncclFloat, ncclSum, g_comm, st));
    NNTK_HIP_TRY(hipMemcpyAsync(block, g_stage, n * sizeof(float), hipMemcpyDeviceToHost, st));
    NNTK_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int nntk_shim_dist_barrier(void) {
    std::lock_guard<std::mutex> lk(g_mutex);
    if (!g_comm) return 0;
    if (g_stage_n < 1) { NNTK_HIP_TRY(hipMalloc((void **)&g_stage, 64)); g_stage_n = 16; }
    hipStream_t st = nntk_stream();
    NNTK_HIP_TRY(hipMemsetAsync(g_stage, 0, sizeof(float), st));
    NNTK_RCCL_TRY(R.AllReduce(g_stage, g_stage, 1, ncclFloat, ncclSum, g_comm, st));
    NNTK_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int nntk_shim_dist_finalize(void) {
    std::lock_guard<std::mutex> lk(g_mutex);
    if (g_comm) { NNTK_RCCL_TRY(R.CommDestroy(g_comm)); g_comm = nullptr; }
    if (g_stage) { (void)hipFree(g_stage); g_stage = nullptr; g_stage_n = 0; }
    g_rank = 0; g_world = 1;
    return 0;
}

}  // extern "C"
