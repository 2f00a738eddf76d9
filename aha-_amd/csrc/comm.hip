// Score all-gather over RCCL: the only collective of the path.  Streams are independent, so ranks exchange
// nothing but their per-step score rows (fp32 [rows][3], a few hundred bytes): one ncclAllGather on the caller's HIP stream,
// no reduction.  The communicator is built from a unique id that the host side distributes however it likes (the Python host
// uses torch.distributed's store); nothing here depends on torch.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "../../include/aha_amd.h"

struct aha_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    std::string err;
};

static thread_local std::string g_comm_err;

extern "C" const char* aha_comm_last_error(void) { return g_comm_err.c_str(); }

extern "C" int aha_comm_unique_id(void* id_out, size_t bytes) {
    if (!id_out || bytes < sizeof(ncclUniqueId)) { g_comm_err = "id buffer must hold AHA_COMM_ID_BYTES"; return -22; }
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { g_comm_err = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r); return -5; }
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int aha_comm_init_rank(const void* id_in, size_t bytes, int nranks, int rank, int device, aha_comm** out) {
    if (!id_in || !out || bytes < sizeof(ncclUniqueId) || nranks <= 0 || rank < 0 || rank >= nranks) { g_comm_err = "bad argument"; return -22; }
    if (hipSetDevice(device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return -5; }
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    aha_comm* c = new aha_comm();
    c->nranks = nranks; c->rank = rank; c->device = device;
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { g_comm_err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); delete c; return -5; }
    *out = c;
    return 0;
}

extern "C" int aha_comm_size(const aha_comm* c) { return c ? c->nranks : -22; }
extern "C" int aha_comm_rank(const aha_comm* c) { return c ? c->rank : -22; }

// local: fp32 [rows][3] of this rank; global: fp32 [nranks][rows][3], rank-major (every rank passes the same `rows`; ragged
// ranks pad, aha_amd.sharding).  Asynchronous on `st`.
extern "C" int aha_allgather_scores(aha_comm* c, const float* local, int rows, float* global, aha_hip_stream st) {
    if (!c || !local || !global || rows <= 0) { g_comm_err = "bad argument"; return -22; }
    ncclResult_t r = ncclAllGather(local, global, (size_t)rows * 3, ncclFloat, c->comm, (hipStream_t)st);
    if (r != ncclSuccess) { g_comm_err = std::string("ncclAllGather: ") + ncclGetErrorString(r); return -5; }
    return 0;
}

extern "C" void aha_comm_destroy(aha_comm* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
}
