// The exchange step of the multi-GPU split (include/mirt.h: mirt_reduce, mirt_comm_*): every rank has rendered
// its own range of a pixel's samples into its float4 accumulation buffer, and one RCCL reduce over xGMI adds the
// buffers onto the root rank, on each context's own stream. The reference hides its parallelism inside draw()
// (rayon row tasks, renderer.rs:69-72); the sample-range split plus this sum is what stands there when the
// workers are GPUs.
//
// librccl is loaded with dlopen() on first use, not linked: a single-GPU caller never pays for loading it, and a
// process that already holds an RCCL (one that imported torch, say) gets that same copy, since both carry the
// name librccl.so.1.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "mirt_ctx.h"

static_assert(MIRT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id handed between ranks is RCCL's ncclUniqueId");

namespace {

struct Rccl {
    void* handle = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

// Loads librccl once per process; nullptr (with the reason in *why) when it cannot be loaded.
const Rccl* rccl(std::string* why) {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* e = dlerror();
            r.error = std::string("librccl could not be loaded: ") + (e ? e : "unknown error");
            return;
        }
        auto sym = [](const char* name, auto& fn) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r.handle, name));
            if (!fn && r.error.empty()) r.error = std::string("librccl lacks ") + name;
        };
        sym("ncclGetUniqueId", r.GetUniqueId);
        sym("ncclCommInitRank", r.CommInitRank);
        sym("ncclCommInitAll", r.CommInitAll);
        sym("ncclCommDestroy", r.CommDestroy);
        sym("ncclGetErrorString", r.GetErrorString);
        sym("ncclReduce", r.Reduce);
        sym("ncclGroupStart", r.GroupStart);
        sym("ncclGroupEnd", r.GroupEnd);
    });
    if (!r.error.empty()) {
        if (why) *why = r.error;
        return nullptr;
    }
    return &r;
}

int comm_fail(mirt_ctx* c, const Rccl* r, const char* what, ncclResult_t rc) {
    return mirt::ctx_fail(c, MIRT_ERR_COMM, std::string(what) + ": " + (r ? r->GetErrorString(rc) : "RCCL unavailable"));
}

std::atomic<unsigned long long> g_groups{0};

}  // namespace

void mirt::comm_release(mirt_ctx* c) {
    if (!c || !c->comm) return;
    if (const Rccl* r = rccl(nullptr)) (void)r->CommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    c->comm_ranks = 0;
    c->comm_rank = -1;
    c->comm_group = 0;
}

extern "C" {

int mirt_comm_unique_id(uint8_t id[MIRT_COMM_ID_BYTES]) {
    if (!id) return MIRT_ERR_INVALID;
    std::string why;
    const Rccl* r = rccl(&why);
    if (!r) return mirt::ctx_fail(nullptr, MIRT_ERR_COMM, "mirt_comm_unique_id: " + why);
    ncclUniqueId u;
    ncclResult_t rc = r->GetUniqueId(&u);
    if (rc != ncclSuccess) return comm_fail(nullptr, r, "ncclGetUniqueId", rc);
    std::memcpy(id, u.internal, MIRT_COMM_ID_BYTES);
    return MIRT_OK;
}

int mirt_comm_init_rank(mirt_ctx* c, int n_ranks, int rank, const uint8_t id[MIRT_COMM_ID_BYTES]) {
    if (!c) return MIRT_ERR_INVALID;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return mirt::ctx_fail(c, MIRT_ERR_INVALID, "mirt_comm_init_rank: bad rank / id");
    std::string why;
    const Rccl* r = rccl(&why);
    if (!r) return mirt::ctx_fail(c, MIRT_ERR_COMM, "mirt_comm_init_rank: " + why);
    if (hipSetDevice(c->device) != hipSuccess) return mirt::ctx_fail(c, MIRT_ERR_HIP, "mirt_comm_init_rank: hipSetDevice failed");
    mirt::comm_release(c);
    ncclUniqueId u;
    std::memcpy(u.internal, id, MIRT_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t rc = r->CommInitRank(&comm, n_ranks, u, rank);
    if (rc != ncclSuccess) return comm_fail(c, r, "ncclCommInitRank", rc);
    c->comm = comm;
    c->comm_ranks = n_ranks;
    c->comm_rank = rank;
    c->comm_group = 0;
    return MIRT_OK;
}

int mirt_comm_reduce(mirt_ctx* c, int root) {
    if (!c) return MIRT_ERR_INVALID;
    if (!c->comm) return mirt::ctx_fail(c, MIRT_ERR_STATE, "mirt_comm_reduce: mirt_comm_init_rank has not been called");
    if (root < 0 || root >= c->comm_ranks) return mirt::ctx_fail(c, MIRT_ERR_INVALID, "mirt_comm_reduce: root out of range");
    if (!c->accum) return mirt::ctx_fail(c, MIRT_ERR_STATE, "mirt_comm_reduce: no frame begun");
    std::string why;
    const Rccl* r = rccl(&why);
    if (!r) return mirt::ctx_fail(c, MIRT_ERR_COMM, "mirt_comm_reduce: " + why);
    if (hipSetDevice(c->device) != hipSuccess) return mirt::ctx_fail(c, MIRT_ERR_HIP, "mirt_comm_reduce: hipSetDevice failed");
    const size_t count = (size_t)c->width * c->height * 4;
    // in place: the root's buffer receives the sum, the other ranks' buffers are only read
    (void)hipEventRecord(c->ev_r0, c->stream);
    ncclResult_t rc = r->Reduce(c->accum, c->accum, count, ncclFloat, ncclSum, root, static_cast<ncclComm_t>(c->comm), c->stream);
    if (rc != ncclSuccess) return comm_fail(c, r, "ncclReduce", rc);
    c->reduce_timed = hipEventRecord(c->ev_r1, c->stream) == hipSuccess;
    return MIRT_OK;
}

int mirt_comm_ranks(const mirt_ctx* c) { return c ? c->comm_ranks : 0; }

int mirt_reduce(mirt_ctx** ctxs, int n, int root) {
    if (!ctxs || n < 1) return MIRT_ERR_INVALID;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return MIRT_ERR_INVALID;
    mirt_ctx* c0 = ctxs[0];
    if (root < 0 || root >= n) return mirt::ctx_fail(c0, MIRT_ERR_INVALID, "mirt_reduce: root out of range");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]->accum) return mirt::ctx_fail(c0, MIRT_ERR_STATE, "mirt_reduce: context " + std::to_string(i) + " has no frame begun");
        if (ctxs[i]->width != c0->width || ctxs[i]->height != c0->height)
            return mirt::ctx_fail(c0, MIRT_ERR_INVALID, "mirt_reduce: the contexts' frames differ in size");
        for (int j = 0; j < i; ++j)
            if (ctxs[j]->device == ctxs[i]->device) return mirt::ctx_fail(c0, MIRT_ERR_INVALID, "mirt_reduce: two contexts share a device");
    }
    std::string why;
    const Rccl* r = rccl(&why);
    if (!r) return mirt::ctx_fail(c0, MIRT_ERR_COMM, "mirt_reduce: " + why);
    // The communicators of a set of contexts are made on its first reduce and kept: the same contexts, in the same
    // order, find them again through their shared group number.
    bool have = c0->comm && c0->comm_group != 0 && c0->comm_ranks == n;
    for (int i = 0; have && i < n; ++i)
        have = ctxs[i]->comm && ctxs[i]->comm_group == c0->comm_group && ctxs[i]->comm_rank == i && ctxs[i]->comm_ranks == n;
    if (!have) {
        std::vector<ncclComm_t> comms(n);
        std::vector<int> devs(n);
        for (int i = 0; i < n; ++i) {
            mirt::comm_release(ctxs[i]);
            devs[i] = ctxs[i]->device;
        }
        ncclResult_t rc = r->CommInitAll(comms.data(), n, devs.data());
        if (rc != ncclSuccess) return comm_fail(c0, r, "ncclCommInitAll", rc);
        const unsigned long long group = ++g_groups;
        for (int i = 0; i < n; ++i) {
            ctxs[i]->comm = comms[i];
            ctxs[i]->comm_ranks = n;
            ctxs[i]->comm_rank = i;
            ctxs[i]->comm_group = group;
        }
    }
    const size_t count = (size_t)c0->width * c0->height * 4;
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(ctxs[i]->device);
        (void)hipEventRecord(ctxs[i]->ev_r0, ctxs[i]->stream);
    }
    ncclResult_t rc = r->GroupStart();
    if (rc != ncclSuccess) return comm_fail(c0, r, "ncclGroupStart", rc);
    ncclResult_t first_bad = ncclSuccess;
    for (int i = 0; i < n; ++i) {
        rc = r->Reduce(ctxs[i]->accum, ctxs[i]->accum, count, ncclFloat, ncclSum, root, static_cast<ncclComm_t>(ctxs[i]->comm), ctxs[i]->stream);
        if (rc != ncclSuccess && first_bad == ncclSuccess) first_bad = rc;
    }
    rc = r->GroupEnd();
    if (first_bad != ncclSuccess) return comm_fail(c0, r, "ncclReduce", first_bad);
    if (rc != ncclSuccess) return comm_fail(c0, r, "ncclGroupEnd", rc);
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(ctxs[i]->device);
        ctxs[i]->reduce_timed = hipEventRecord(ctxs[i]->ev_r1, ctxs[i]->stream) == hipSuccess;
    }
    return MIRT_OK;
}

}  // extern "C"
