// Data-parallel exchange step of the tracking path: the ONE collective of a job -- an RCCL
// ncclAllGather of every rank's relative poses (4x4 f64 + a status word = 17 doubles per frame pair) over xGMI.
// Counterpart in the reference: none (DF-VO is single-process); what it replaces is the per-frame
// update_global_pose hand-over of /root/reference/libs/dfvo.py:109-119,157-161 between the chunks of a sequence.
//
// RCCL is bound at first use (dlopen of librccl.so.1 -- the copy the process already holds when torch is loaded, else the
// ROCm one), so that libdfvo_hip.so itself loads on hosts that never run more than one rank.  No fallback: a missing
// library or a failed collective is an error.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

#include <cstdlib>
#include <cstring>

#include "dfvo_common.h"

namespace dfvo {
namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // DFVO_RCCL_LIB (test hook, tests/test_dist_cpu.py): the one library name to try instead of the list below
        const char* forced = getenv("DFVO_RCCL_LIB");
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        std::string why;
        for (const char* n : names) {
            const char* name = forced ? forced : n;
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
            const char* e = dlerror();  // ONE call: dlerror() returns the message once and clears it
            why = std::string("dlopen(") + name + "): " + (e ? e : "not found");
            if (forced) break;
        }
        if (!r.h) {
            r.err = why;
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(r.h, s);
            if (!p && r.err.empty()) r.err = std::string("librccl: missing symbol ") + s;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}
}  // namespace
}  // namespace dfvo

struct dfvo_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    int device = -1;  // the HIP device that was current at dfvo_comm_create: every collective must be issued on it
    hipStream_t stream = nullptr;
    double *d_send = nullptr, *d_recv = nullptr;
    size_t cap_rows = 0;  // rows per rank the staging buffers hold
};

#define DFVO_NCCL_CHECK(expr)                                                                                   \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) {                                                                                \
            dfvo::set_last_error(std::string(#expr) + ": " + (R->GetErrorString ? R->GetErrorString(_r) : "?")); \
            return DFVO_ERR_HIP;                                                                                \
        }                                                                                                       \
    } while (0)

extern "C" {

int dfvo_comm_unique_id(uint8_t* h_id128) {
    DFVO_ARG_CHECK(h_id128, "dfvo_comm_unique_id: null output");
    dfvo::Rccl* R = dfvo::rccl();
    DFVO_ARG_CHECK(R->err.empty(), R->err.c_str());
    static_assert(sizeof(ncclUniqueId) == DFVO_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    DFVO_NCCL_CHECK(R->GetUniqueId(&id));
    memcpy(h_id128, &id, sizeof(id));
    return DFVO_OK;
}

int dfvo_comm_create(const uint8_t* h_id128, int world, int rank, dfvo_comm** out) {
    DFVO_ARG_CHECK(h_id128 && out && world >= 1 && rank >= 0 && rank < world, "dfvo_comm_create: bad argument");
    dfvo::Rccl* R = dfvo::rccl();
    DFVO_ARG_CHECK(R->err.empty(), R->err.c_str());
    dfvo_comm* c = new dfvo_comm();
    c->world = world;
    c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) {
        dfvo::set_last_error("dfvo_comm_create: no current HIP device (dfvo_set_device first)");
        delete c;
        return DFVO_ERR_HIP;
    }
    ncclUniqueId id;
    memcpy(&id, h_id128, sizeof(id));
    ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        dfvo::set_last_error(std::string("ncclCommInitRank: ") + R->GetErrorString(r));
        delete c;
        return DFVO_ERR_HIP;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        dfvo::set_last_error("dfvo_comm_create: hipStreamCreate failed");
        R->CommDestroy(c->comm);
        delete c;
        return DFVO_ERR_HIP;
    }
    *out = c;
    return DFVO_OK;
}

int dfvo_comm_destroy(dfvo_comm* c) {
    if (!c) return DFVO_OK;
    dfvo::Rccl* R = dfvo::rccl();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && R->CommDestroy) (void)R->CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return DFVO_OK;
}

// the collective proper: every rank contributes rows_per_rank x 17 doubles (its own rows first, the rest padding)
int dfvo_allgather_poses_device(dfvo_comm* c, const double* d_send, int rows_per_rank, double* d_recv, void* stream) {
    DFVO_ARG_CHECK(c && d_send && d_recv && rows_per_rank > 0, "dfvo_allgather_poses_device: bad argument");
    dfvo::Rccl* R = dfvo::rccl();
    int dev = -1;
    DFVO_HIP_CHECK(hipGetDevice(&dev));
    DFVO_ARG_CHECK(dev == c->device, "dfvo_allgather_poses: the current device is not the one the communicator was created on "
                                     "(a host thread that forgot dfvo_set_device would put two ranks on one GPU)");
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    DFVO_NCCL_CHECK(R->AllGather(d_send, d_recv, (size_t)rows_per_rank * DFVO_POSE_ROW, ncclDouble, c->comm, s));
    return DFVO_OK;
}

int dfvo_allgather_poses(dfvo_comm* c, const double* h_rows, int n_local, const int* counts, double* h_out) {
    DFVO_ARG_CHECK(c && counts && h_out && n_local >= 0 && (h_rows || n_local == 0), "dfvo_allgather_poses: bad argument");
    DFVO_ARG_CHECK(counts[c->rank] == n_local, "dfvo_allgather_poses: counts[rank] != n_local");
    int nmax = 0;
    for (int r = 0; r < c->world; ++r) {
        DFVO_ARG_CHECK(counts[r] >= 0, "dfvo_allgather_poses: negative count");
        nmax = counts[r] > nmax ? counts[r] : nmax;
    }
    if (nmax == 0) return DFVO_OK;
    if ((size_t)nmax > c->cap_rows) {
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr;
        c->cap_rows = 0;
        DFVO_HIP_CHECK(hipMalloc(&c->d_send, (size_t)nmax * DFVO_POSE_ROW * sizeof(double)));
        DFVO_HIP_CHECK(hipMalloc(&c->d_recv, (size_t)nmax * DFVO_POSE_ROW * sizeof(double) * c->world));
        c->cap_rows = (size_t)nmax;
    }
    const size_t row_b = DFVO_POSE_ROW * sizeof(double);
    DFVO_HIP_CHECK(hipMemsetAsync(c->d_send, 0, (size_t)nmax * row_b, c->stream));
    if (n_local) DFVO_HIP_CHECK(hipMemcpyAsync(c->d_send, h_rows, (size_t)n_local * row_b, hipMemcpyHostToDevice, c->stream));
    const int rc = dfvo_allgather_poses_device(c, c->d_send, nmax, c->d_recv, c->stream);
    if (rc != DFVO_OK) return rc;
    size_t off = 0;
    for (int r = 0; r < c->world; ++r) {  // trim the padding: rows of rank r start at r * nmax
        if (counts[r])
            DFVO_HIP_CHECK(hipMemcpyAsync(h_out + off * DFVO_POSE_ROW, c->d_recv + (size_t)r * nmax * DFVO_POSE_ROW,
                                          (size_t)counts[r] * row_b, hipMemcpyDeviceToHost, c->stream));
        off += (size_t)counts[r];
    }
    DFVO_HIP_CHECK(hipStreamSynchronize(c->stream));
    return DFVO_OK;
}

}  // extern "C"
