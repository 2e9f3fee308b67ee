// Native implementations of `ts_comm` (include/tapstark.h), so that a Rust / C host has something to
// link for ts_prove_sharded without Python (SURVEY.md section 8(e) "Collectives"; the reference has
// no communication layer at all):
//
//   * RCCL over xGMI, one process (or thread) per GPU: ncclAllGather / ncclBroadcast enqueued on the
//     context's own HIP stream -- no host synchronisation per collective.  librccl is bound at run
//     time (dlopen), preferring a copy the process already holds (a PyTorch-ROCm process carries its
//     own), so that the library itself keeps linking against the HIP runtime only.  Bootstrap is the
//     caller's: rank 0 makes a 128-byte unique id (ts_rccl_unique_id) and hands it to its peers by
//     whatever channel it has (a file, a socket, torch.distributed, MPI).
//   * an in-process group: G ranks = G host threads of ONE process, each with its own context (all
//     on one device, or one device each with peer copies over xGMI).  A collective is a rendezvous
//     of the G threads plus device-to-device copies on the receiver's stream.  This is what the
//     tests use to run 8 ranks on a one-GPU box, where 8 processes may not share the card.
//
// Failure handling: a rank that fails after the first collective must not leave its peers waiting
// for ever.  ts_comm.abort (optional) is called by ts_prove_sharded on the failing rank: the local
// group is poisoned (every waiting or later rendezvous returns an error), the RCCL communicator is
// aborted (ncclCommAbort makes pending collectives on the peers fail rather than hang).
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "../../include/tapstark.h"
#include "abi_types.hpp"
#include "context.hpp"

// ------------------------------------------------------------------ in-process group
struct ts_comm_group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool poisoned = false;
    // per collective: what every rank published
    std::vector<const void*> send;
    std::vector<void*> recv;
    std::vector<int> device;
    struct Slot {
        ts_comm_group* g;
        int rank;
    };
    std::vector<Slot> slots;

    int timeout_s = 600;  // TS_COMM_TIMEOUT_S / ts_comm_local_group_set_timeout

    // all ranks arrive; returns false if the group was poisoned (or on a timeout -- 10 minutes unless
    // configured -- which poisons it: a peer died without telling anyone)
    bool rendezvous() {
        std::unique_lock<std::mutex> lk(mu);
        if (poisoned) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return true;
        }
        const bool ok = cv.wait_for(lk, std::chrono::seconds(timeout_s),
                                    [&] { return generation != gen || poisoned; });
        if (!ok) {
            poisoned = true;
            cv.notify_all();
        }
        return !poisoned;
    }
    void poison() {
        std::lock_guard<std::mutex> lk(mu);
        poisoned = true;
        cv.notify_all();
    }
};

namespace {

int local_all_gather(void* user, const void* send_dev, void* recv_dev, size_t bytes, void* hip_stream) {
    auto* s = static_cast<ts_comm_group::Slot*>(user);
    ts_comm_group* g = s->g;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    // my contribution must be complete before a peer reads it
    if (hipStreamSynchronize(stream) != hipSuccess) {
        g->poison();
        return 1;
    }
    g->send[s->rank] = send_dev;
    g->recv[s->rank] = recv_dev;
    g->device[s->rank] = dev;
    if (!g->rendezvous()) return 1;
    bool ok = true;
    for (int r = 0; r < g->world && ok; r++) {
        void* dst = static_cast<char*>(recv_dev) + (size_t)r * bytes;
        hipError_t e = g->device[r] == dev
                           ? hipMemcpyAsync(dst, g->send[r], bytes, hipMemcpyDeviceToDevice, stream)
                           : hipMemcpyPeerAsync(dst, dev, g->send[r], g->device[r], bytes, stream);
        ok = e == hipSuccess;
    }
    // the peers' send buffers may be reused as soon as this returns on THEIR side: finish reading
    ok = ok && hipStreamSynchronize(stream) == hipSuccess;
    if (!ok) g->poison();
    if (!g->rendezvous()) return 1;
    return ok ? 0 : 1;
}

int local_broadcast(void* user, void* buf_dev, size_t bytes, int root, void* hip_stream) {
    auto* s = static_cast<ts_comm_group::Slot*>(user);
    ts_comm_group* g = s->g;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || root < 0 || root >= g->world) return 1;
    if (hipStreamSynchronize(stream) != hipSuccess) {
        g->poison();
        return 1;
    }
    g->send[s->rank] = buf_dev;
    g->device[s->rank] = dev;
    if (!g->rendezvous()) return 1;
    bool ok = true;
    if (s->rank != root) {
        hipError_t e = g->device[root] == dev
                           ? hipMemcpyAsync(buf_dev, g->send[root], bytes, hipMemcpyDeviceToDevice, stream)
                           : hipMemcpyPeerAsync(buf_dev, dev, g->send[root], g->device[root], bytes, stream);
        ok = e == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    }
    if (!ok) g->poison();
    if (!g->rendezvous()) return 1;
    return ok ? 0 : 1;
}

void local_abort(void* user) { static_cast<ts_comm_group::Slot*>(user)->g->poison(); }

// ------------------------------------------------------------------ RCCL
struct NcclUniqueId {
    char internal[128];
};
typedef void* NcclComm;
enum { NCCL_UINT8 = 1 };  // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1

struct Rccl {
    void* lib = nullptr;
    int (*get_unique_id)(NcclUniqueId*) = nullptr;
    int (*comm_init_rank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*comm_destroy)(NcclComm) = nullptr;
    int (*comm_abort)(NcclComm) = nullptr;
    int (*comm_count)(NcclComm, int*) = nullptr;
    int (*comm_user_rank)(NcclComm, int*) = nullptr;
    int (*comm_cu_device)(NcclComm, int*) = nullptr;
    int (*get_version)(int*) = nullptr;
    const char* (*get_error_string)(int) = nullptr;
    bool ok = false;
};

Rccl load_rccl() {
    Rccl r;
    // a copy already mapped into the process first (PyTorch-ROCm ships its own librccl.so and two
    // RCCL instances in one process would each bring their own kernels and topology state)
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                                 "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
    if (!r.lib) return r;
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.lib, "ncclCommInitRank");
    r.all_gather = (decltype(r.all_gather))dlsym(r.lib, "ncclAllGather");
    r.broadcast = (decltype(r.broadcast))dlsym(r.lib, "ncclBroadcast");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.lib, "ncclCommDestroy");
    r.comm_abort = (decltype(r.comm_abort))dlsym(r.lib, "ncclCommAbort");
    r.comm_count = (decltype(r.comm_count))dlsym(r.lib, "ncclCommCount");
    r.comm_user_rank = (decltype(r.comm_user_rank))dlsym(r.lib, "ncclCommUserRank");
    r.comm_cu_device = (decltype(r.comm_cu_device))dlsym(r.lib, "ncclCommCuDevice");
    r.get_version = (decltype(r.get_version))dlsym(r.lib, "ncclGetVersion");
    r.get_error_string = (decltype(r.get_error_string))dlsym(r.lib, "ncclGetErrorString");
    r.ok = r.get_unique_id && r.comm_init_rank && r.all_gather && r.broadcast && r.comm_destroy;
    return r;
}
Rccl& rccl() {
    static Rccl r = load_rccl();
    return r;
}

}  // namespace

struct ts_rccl_comm {
    NcclComm comm = nullptr;
    int rank = 0, world = 1;
    std::atomic<bool> aborted{false};
    bool checked = false;  // ncclCommCount / ncclCommUserRank confirmed rank and world at creation
};

namespace {

int rccl_all_gather(void* user, const void* send_dev, void* recv_dev, size_t bytes, void* hip_stream) {
    auto* c = static_cast<ts_rccl_comm*>(user);
    if (c->aborted) return 1;
    return rccl().all_gather(send_dev, recv_dev, bytes, NCCL_UINT8, c->comm,
                             static_cast<hipStream_t>(hip_stream)) == 0
               ? 0
               : 1;
}
int rccl_broadcast(void* user, void* buf_dev, size_t bytes, int root, void* hip_stream) {
    auto* c = static_cast<ts_rccl_comm*>(user);
    if (c->aborted) return 1;
    return rccl().broadcast(buf_dev, buf_dev, bytes, NCCL_UINT8, root, c->comm,
                            static_cast<hipStream_t>(hip_stream)) == 0
               ? 0
               : 1;
}
void rccl_abort(void* user) {
    auto* c = static_cast<ts_rccl_comm*>(user);
    if (!c->aborted.exchange(true) && rccl().comm_abort && c->comm) {
        rccl().comm_abort(c->comm);
        c->comm = nullptr;
    }
}

}  // namespace

extern "C" {

ts_status ts_comm_local_group_create(int world, ts_comm_group** out) {
    if (!out || world < 1 || world > 64) return TS_ERR_INVALID;
    auto* g = new (std::nothrow) ts_comm_group();
    if (!g) return TS_ERR_OOM;
    g->world = world;
    if (const char* e = getenv("TS_COMM_TIMEOUT_S")) {
        const int t = atoi(e);
        if (t > 0) g->timeout_s = t;
    }
    g->send.assign(world, nullptr);
    g->recv.assign(world, nullptr);
    g->device.assign(world, 0);
    g->slots.resize(world);
    for (int r = 0; r < world; r++) g->slots[r] = ts_comm_group::Slot{g, r};
    *out = g;
    return TS_OK;
}

ts_status ts_comm_local_get(ts_comm_group* group, int rank, ts_comm* out) {
    if (!group || !out || rank < 0 || rank >= group->world) return TS_ERR_INVALID;
    memset(out, 0, sizeof *out);
    out->rank = rank;
    out->world = group->world;
    out->user = &group->slots[rank];
    out->all_gather = local_all_gather;
    out->broadcast = local_broadcast;
    out->abort = local_abort;
    return TS_OK;
}

void ts_comm_local_group_destroy(ts_comm_group* group) { delete group; }

ts_status ts_comm_local_group_set_timeout(ts_comm_group* group, int seconds) {
    if (!group || seconds < 1) return TS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(group->mu);
    group->timeout_s = seconds;
    return TS_OK;
}

ts_status ts_comm_local_group_reset(ts_comm_group* group) {
    if (!group) return TS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(group->mu);
    group->poisoned = false;
    group->arrived = 0;
    group->generation++;  // a straggler of the aborted collective (there must be none) would not match
    return TS_OK;
}

int ts_rccl_available(void) { return rccl().ok ? 1 : 0; }

ts_status ts_rccl_unique_id(uint8_t out[128]) {
    if (!out) return TS_ERR_INVALID;
    if (!rccl().ok) return TS_ERR_UNSUPPORTED;
    NcclUniqueId id;
    if (rccl().get_unique_id(&id) != 0) return TS_ERR_COMM;
    memcpy(out, id.internal, 128);
    return TS_OK;
}

ts_status ts_comm_rccl_create(ts_ctx* ctx, const uint8_t unique_id[128], int rank, int world,
                              ts_comm* out, ts_rccl_comm** handle) {
    if (!ctx || !unique_id || !out || !handle || world < 1 || rank < 0 || rank >= world)
        return TS_ERR_INVALID;
    *handle = nullptr;
    if (!rccl().ok) return TS_ERR_UNSUPPORTED;
    // the communicator binds to the current device: make it the context's
    if (ts_ctx_synchronize(ctx) != TS_OK) return TS_ERR_HIP;
    auto* c = new (std::nothrow) ts_rccl_comm();
    if (!c) return TS_ERR_OOM;
    NcclUniqueId id;
    memcpy(id.internal, unique_id, 128);
    if (rccl().comm_init_rank(&c->comm, world, id, rank) != 0) {
        delete c;
        ctx->ctx.last_error = "ncclCommInitRank failed";
        return TS_ERR_COMM;
    }
    c->rank = rank;
    c->world = world;
    // what RCCL itself believes must be what the caller asked for: a communicator that came up with
    // another size or rank would exchange the wrong slabs without any error
    if (rccl().comm_count && rccl().comm_user_rank) {
        int n = -1, r = -1;
        const bool asked = rccl().comm_count(c->comm, &n) == 0 && rccl().comm_user_rank(c->comm, &r) == 0;
        if (!asked || n != world || r != rank) {
            char msg[200];
            snprintf(msg, sizeof msg,
                     "RCCL communicator mismatch: asked for rank %d of %d, ncclCommUserRank = %d, ncclCommCount = %d%s",
                     rank, world, r, n, asked ? "" : " (query failed)");
            ctx->ctx.last_error = msg;
            // abort so that the peers' first collective fails instead of blocking; a librccl without
            // ncclCommAbort still gets its communicator destroyed rather than leaked
            if (rccl().comm_abort) rccl().comm_abort(c->comm);
            else if (rccl().comm_destroy) rccl().comm_destroy(c->comm);
            delete c;
            return TS_ERR_COMM;
        }
        c->checked = true;
    }
    memset(out, 0, sizeof *out);
    out->rank = rank;
    out->world = world;
    out->user = c;
    out->all_gather = rccl_all_gather;
    out->broadcast = rccl_broadcast;
    out->abort = rccl_abort;
    *handle = c;
    return TS_OK;
}

ts_status ts_comm_rccl_info(const ts_rccl_comm* c, ts_rccl_info* out) {
    if (!c || !out) return TS_ERR_INVALID;
    memset(out, 0, sizeof *out);
    out->comm_count = out->comm_user_rank = out->comm_device = out->rccl_version = -1;
    out->rank = c->rank;
    out->world = c->world;
    out->aborted = c->aborted ? 1 : 0;
    out->checked = c->checked ? 1 : 0;
    if (rccl().get_version) rccl().get_version(&out->rccl_version);
    if (!c->comm || c->aborted) return TS_OK;
    if (rccl().comm_count) rccl().comm_count(c->comm, &out->comm_count);
    if (rccl().comm_user_rank) rccl().comm_user_rank(c->comm, &out->comm_user_rank);
    if (rccl().comm_cu_device) rccl().comm_cu_device(c->comm, &out->comm_device);
    return TS_OK;
}

void ts_comm_rccl_destroy(ts_rccl_comm* c) {
    if (!c) return;
    if (c->comm && !c->aborted) rccl().comm_destroy(c->comm);
    delete c;
}

}  // extern "C"
