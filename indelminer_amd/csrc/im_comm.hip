// im_comm.hip -- the collectives of the multi-GPU design over RCCL / xGMI: the all-gather of the
// per-rank logs, and -- only when pieces of one contig were walked by several ranks -- the sum of the ranks' shares of the
// depth array.
//
// The reference has no communication at all (its only parallel mode is running
// several processes with -c regions, src/indelminer.c:536-542,711-713); contigs are
// independent, so the shards exchange nothing until each holds its cluster list.
// librccl is loaded lazily (dlopen) so that single-GPU users never touch it.

#include "im_device.hpp"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>

struct im_comm {
    im_ctx* ctx;
    ncclComm_t comm;
    int rank, world;
};

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char err[256] = {0};
} g_rccl;

// The librccl to load is the one that belongs to the HIP runtime THIS library is linked with (found through
// dladdr of a HIP entry point): a process that has imported torch for its control plane already carries
// torch's own bundled librccl + libamdhip64 under the same sonames, and a communicator made by that copy would
// run on a second HIP runtime that knows none of our allocations.  RTLD_LOCAL | RTLD_DEEPBIND keeps the copy
// we load bound to its own symbols.
bool load_rccl()
{
    if (g_rccl.handle) return true;
    char beside_hip[1024] = "";
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
        const char* slash = strrchr(info.dli_fname, '/');
        if (slash && (size_t)(slash - info.dli_fname) + 16 < sizeof beside_hip) {
            memcpy(beside_hip, info.dli_fname, (size_t)(slash - info.dli_fname) + 1);
            strcat(beside_hip, "librccl.so.1");
        }
    }
    char from_env[1024] = "";
    if (const char* rp = getenv("ROCM_PATH")) snprintf(from_env, sizeof from_env, "%s/lib/librccl.so.1", rp);
    const char* names[] = { beside_hip, from_env, "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so" };
    for (const char* n : names) {
        if (!n[0]) continue;
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) { snprintf(g_rccl.err, sizeof g_rccl.err, "cannot load librccl: %s", dlerror()); return false; }
#define LOAD(field, sym) \
    *(void**)(&g_rccl.field) = dlsym(g_rccl.handle, sym); \
    if (!g_rccl.field) { snprintf(g_rccl.err, sizeof g_rccl.err, "librccl lacks %s", sym); return false; }
    LOAD(GetUniqueId, "ncclGetUniqueId")
    LOAD(CommInitRank, "ncclCommInitRank")
    LOAD(AllGather, "ncclAllGather")
    LOAD(AllReduce, "ncclAllReduce")
    LOAD(Send, "ncclSend")
    LOAD(Recv, "ncclRecv")
    LOAD(GroupStart, "ncclGroupStart")
    LOAD(GroupEnd, "ncclGroupEnd")
    LOAD(CommDestroy, "ncclCommDestroy")
    LOAD(GetErrorString, "ncclGetErrorString")
#undef LOAD
    return true;
}

}  // namespace

extern "C" {

int im_comm_unique_id(void* id_bytes)
{
    if (!id_bytes) return IM_E_ARG;
    if (!load_rccl()) return IM_E_HIP;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return IM_E_HIP;
    memcpy(id_bytes, &id, IM_COMM_ID_BYTES);
    return IM_OK;
}

const char* im_comm_last_error(void) { return g_rccl.err; }

int im_comm_init(im_ctx* ctx, const void* id_bytes, int rank, int world, im_comm** out)
{
    if (!ctx || !id_bytes || !out || world < 1 || rank < 0 || rank >= world) return IM_E_ARG;
    if (!load_rccl()) return IM_E_HIP;
    ncclUniqueId id;
    memcpy(&id, id_bytes, IM_COMM_ID_BYTES);
    // the communicator binds to the calling thread's current device: make that the context's
    if (hipSetDevice(im_ctx_device(ctx)) != hipSuccess) {
        snprintf(g_rccl.err, sizeof g_rccl.err, "hipSetDevice(%d) failed", im_ctx_device(ctx));
        return IM_E_HIP;
    }
    im_comm* c = new im_comm();
    c->ctx = ctx; c->rank = rank; c->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        snprintf(g_rccl.err, sizeof g_rccl.err, "ncclCommInitRank: %s", g_rccl.GetErrorString(r));
        delete c;
        return IM_E_HIP;
    }
    *out = c;
    return IM_OK;
}

int im_comm_allgather(im_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream)
{
    if (!c) return IM_E_ARG;
    ncclResult_t r = g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, ncclInt8, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) {
        snprintf(g_rccl.err, sizeof g_rccl.err, "ncclAllGather: %s", g_rccl.GetErrorString(r));
        return IM_E_HIP;
    }
    return IM_OK;
}

int im_comm_allreduce_sum_i32(im_comm* c, int32_t* buf_dev, size_t count, void* stream)
{
    if (!c || !buf_dev) return IM_E_ARG;
    ncclResult_t r = g_rccl.AllReduce(buf_dev, buf_dev, count, ncclInt32, ncclSum, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) {
        snprintf(g_rccl.err, sizeof g_rccl.err, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
        return IM_E_HIP;
    }
    return IM_OK;
}

int im_comm_exchange(im_comm* c, int32_t n, const int32_t* dir, const int32_t* peer, void* const* dev, const size_t* bytes, void* stream)
{
    if (!c || n < 0 || (n > 0 && (!dir || !peer || !dev || !bytes))) return IM_E_ARG;
    for (int32_t k = 0; k < n; k++)
        if (peer[k] < 0 || peer[k] >= c->world || (bytes[k] && !dev[k])) return IM_E_ARG;
    if (n == 0) return IM_OK;
    // one group: the operations are matched rank to rank in the order they are listed, and none of them can block another
    ncclResult_t r = g_rccl.GroupStart();
    for (int32_t k = 0; r == ncclSuccess && k < n; k++) {
        if (!bytes[k]) continue;
        r = dir[k] == 0 ? g_rccl.Send(dev[k], bytes[k], ncclInt8, peer[k], c->comm, (hipStream_t)stream)
                        : g_rccl.Recv(dev[k], bytes[k], ncclInt8, peer[k], c->comm, (hipStream_t)stream);
    }
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) {
        snprintf(g_rccl.err, sizeof g_rccl.err, "ncclSend / ncclRecv group: %s", g_rccl.GetErrorString(r));
        return IM_E_HIP;
    }
    return IM_OK;
}

void im_comm_destroy(im_comm* c)
{
    if (!c) return;
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
