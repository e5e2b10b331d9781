// comm.hip — vmv_comm_*: the two collectives of the frame-sharded sampler issued from C on the plan's stream.
//
// The reference has no counterpart (its multi-GPU mode is replicas: inference_text2video_entrance.py:79,152-156); SURVEY §8b's last
// row asks for `vmv_comm_*` wrappers around an RCCL communicator injected by the Python host.  A frame-parallel forward has 183
// collectives per CFG branch (2 layout switches per temporal block + one totals gather per all-frame GroupNorm); issued from Python
// between vmv_plan_run_range() calls they cost ~25 us of host time each, which at 8 GPUs (6-8 ms of kernels per rank and step) would
// make the step host-bound.  Recorded as VMV_OP_COMM they are part of the one-call replay (and of a captured hipGraph).
//
// RCCL is NOT a link-time dependency: vmv_comm_load() dlopen()s the librccl the host process already uses (PyTorch ships its own
// copy) and resolves the six entry points below, so libvmv loads — and every non-collective entry point works — without it.
// A "simulated" communicator (vmv_comm_create_sim) stands in for W - 1 absent peers on a single GPU: every collective becomes a
// device-local copy of exactly the bytes the real one would deliver, so the rank-local plan of an 8-GPU run (3 frames, HW / 8 pixels)
// can be built, replayed, timed and profiled on one box (bench.py --simulate-rank).  Its results are NOT a sample (peers' data is
// replaced by this rank's own); its launch sequence, shapes and local traffic are the real rank's.
#include "common.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <cstring>

struct VmvComm {
    ncclComm_t nccl = nullptr;
    int world = 1, rank = 0;
    bool sim = false;
};

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_api;
std::mutex g_api_mu;

template <class F> bool sym(void* h, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}
}  // namespace

extern "C" int vmv_comm_load(const char* rccl_path) {
    std::lock_guard<std::mutex> lk(g_api_mu);
    if (g_api.handle) return VMV_OK;
    void* h = dlopen(rccl_path && rccl_path[0] ? rccl_path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return VMV_ENULL;
    RcclApi a;
    a.handle = h;
    if (!(sym(h, "ncclGetUniqueId", a.GetUniqueId) && sym(h, "ncclCommInitRank", a.CommInitRank) &&
          sym(h, "ncclCommDestroy", a.CommDestroy) && sym(h, "ncclAllGather", a.AllGather) &&
          sym(h, "ncclAllToAll", a.AllToAll) && sym(h, "ncclGetErrorString", a.GetErrorString))) {
        dlclose(h);
        return VMV_EINVAL;
    }
    g_api = a;
    return VMV_OK;
}

extern "C" int vmv_comm_loaded(void) {
    std::lock_guard<std::mutex> lk(g_api_mu);
    return g_api.handle != nullptr;
}

extern "C" int vmv_comm_unique_id(void* id128) {
    if (!id128) return VMV_ENULL;
    if (!vmv_comm_loaded()) return VMV_EINVAL;
    ncclUniqueId id;
    const ncclResult_t r = g_api.GetUniqueId(&id);
    if (r != ncclSuccess) return VMV_ECOMM;
    static_assert(sizeof(id) == VMV_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return VMV_OK;
}

extern "C" VmvComm* vmv_comm_create(const void* id128, int world, int rank) {
    if (!id128 || world < 1 || rank < 0 || rank >= world || !vmv_comm_loaded()) return nullptr;
    VmvComm* c = new (std::nothrow) VmvComm();
    if (!c) return nullptr;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    if (g_api.CommInitRank(&c->nccl, world, id, rank) != ncclSuccess) {
        delete c;
        return nullptr;
    }
    c->world = world;
    c->rank = rank;
    return c;
}

extern "C" VmvComm* vmv_comm_create_sim(int world, int rank) {
    if (world < 1 || rank < 0 || rank >= world) return nullptr;
    VmvComm* c = new (std::nothrow) VmvComm();
    if (!c) return nullptr;
    c->world = world;
    c->rank = rank;
    c->sim = true;
    return c;
}

extern "C" void vmv_comm_destroy(VmvComm* c) {
    if (!c) return;
    if (c->nccl && vmv_comm_loaded()) g_api.CommDestroy(c->nccl);
    delete c;
}

extern "C" int vmv_comm_world(const VmvComm* c) { return c ? c->world : VMV_ENULL; }
extern "C" int vmv_comm_rank(const VmvComm* c) { return c ? c->rank : VMV_ENULL; }
extern "C" int vmv_comm_is_sim(const VmvComm* c) { return c ? (c->sim ? 1 : 0) : VMV_ENULL; }

extern "C" int vmv_comm_run(const VmvCommParams* p, void* stream) {
    if (!p || !p->comm || !p->send || !p->recv) return VMV_ENULL;
    if (p->bytes <= 0) return VMV_EINVAL;
    const VmvComm* c = p->comm;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t n = (size_t)p->bytes;
    if (c->sim) {
        // absent peers: deliver this rank's own bytes in their place (same byte count into the same buffer) with ONE copy launch,
        // as one collective is one launch: all-to-all recv = send; all-gather recv[j] = send for every j (source stride 0)
        if (p->kind != VMV_COMM_ALL_TO_ALL && p->kind != VMV_COMM_ALL_GATHER) return VMV_EINVAL;
        const bool a2a = p->kind == VMV_COMM_ALL_TO_ALL;
        if ((n & 15) == 0 && vmv_aligned16(p->send) && vmv_aligned16(p->recv) && (n >> 4) * (a2a ? (size_t)c->world : 1) < (1ull << 31)) {
            VmvCopyParams cp = {};
            cp.src = p->send;
            cp.dst = p->recv;
            cp.n0 = a2a ? 1 : c->world;
            cp.n1 = cp.n2 = 1;
            cp.inner16 = (int32_t)((n >> 4) * (a2a ? (size_t)c->world : 1));
            return vmv_permute_copy(&cp, stream);
        }
        for (int r = 0; r < (a2a ? 1 : c->world); ++r) {
            const hipError_t e = hipMemcpyAsync(static_cast<char*>(p->recv) + (size_t)r * n, p->send, a2a ? n * (size_t)c->world : n,
                                                hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return (int)e;
        }
        return VMV_OK;
    }
    if (!c->nccl) return VMV_ENULL;
    ncclResult_t r;
    if (p->kind == VMV_COMM_ALL_TO_ALL) r = g_api.AllToAll(p->send, p->recv, n, ncclInt8, c->nccl, s);
    else if (p->kind == VMV_COMM_ALL_GATHER) r = g_api.AllGather(p->send, p->recv, n, ncclInt8, c->nccl, s);
    else return VMV_EINVAL;
    return r == ncclSuccess ? VMV_OK : VMV_ECOMM;
}
