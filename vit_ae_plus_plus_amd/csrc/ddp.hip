// Data-parallel gradient exchange behind the C ABI: RCCL all-reduce of one gradient bucket on a side HIP stream, forked
// from / joined to the compute stream with events — so the collective can sit INSIDE a captured step graph (one replay
// per optimisation step) instead of being issued by the host between graph replays.
//
// The reference has no gradient exchange at all (its pre-training scripts are single-process per GPU and only all-reduce
// scalars for logging, utils/misc.py:332-340); SURVEY §8(b)/(e) define this boundary for the build: pure data parallelism,
// one SUM all-reduce per gradient bucket, overlapped with the rest of the backward on its own stream.
//
// RCCL is bound at run time (dlopen): the library stays loadable on a box without RCCL, and inside a PyTorch process it
// picks up the librccl.so torch already mapped (one RCCL instance per process).  Rendezvous is the caller's business: rank 0
// calls vitae_ddp_unique_id(), ships the 128 bytes to the other ranks (torch.distributed broadcast, a file, MPI ...), and
// every rank calls vitae_ddp_init().
#include <dlfcn.h>
#include <cstring>
#include "common.hpp"
#include "vitae_hip.h"

namespace {

typedef struct { char internal[128]; } UniqueId;       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
constexpr int NCCL_SUM = 0, NCCL_F32 = 7, NCCL_BF16 = 9;   // rccl.h: ncclSum, ncclFloat32, ncclBfloat16

struct State {
    void* lib = nullptr;
    GetUniqueIdFn get_id = nullptr;
    CommInitRankFn init_rank = nullptr;
    CommDestroyFn destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    Comm comm = nullptr;
    int world = 0, rank = -1;
    static constexpr int NEV = 64;                       // fork / join events, created once at init (nothing allocates later)
    hipEvent_t ev[NEV];
    int next = 0;
    bool events = false;
} g;

bool bind() {
    if (g.lib) return true;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) {
        g.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);       // the instance already in the process (PyTorch's), if any
        if (g.lib) break;
    }
    for (const char* n : names) {
        if (g.lib) break;
        g.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!g.lib) return false;
    g.get_id = (GetUniqueIdFn)dlsym(g.lib, "ncclGetUniqueId");
    g.init_rank = (CommInitRankFn)dlsym(g.lib, "ncclCommInitRank");
    g.destroy = (CommDestroyFn)dlsym(g.lib, "ncclCommDestroy");
    g.all_reduce = (AllReduceFn)dlsym(g.lib, "ncclAllReduce");
    return g.get_id && g.init_rank && g.destroy && g.all_reduce;
}

// Fork / join events come from a ring created at init (nothing allocates later).  Eagerly, and inside a capture, an event is
// recorded and waited for back to back, so re-using ring entries is harmless in itself — but a captured step that needs more than
// the whole ring is a step this file was not sized for (NEV / 2 buckets): refuse it loudly instead of wrapping silently.
unsigned long long g_capture_id = 0;
int g_capture_used = 0;

hipEvent_t next_event(hipStream_t on, bool* ok) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    *ok = true;
    if (on && hipStreamGetCaptureInfo(on, &st, &id) == hipSuccess && st == hipStreamCaptureStatusActive) {
        if (id != g_capture_id) { g_capture_id = id; g_capture_used = 0; }
        if (++g_capture_used > State::NEV) *ok = false;
    }
    hipEvent_t e = g.ev[g.next];
    g.next = (g.next + 1) % State::NEV;
    return e;
}

}  // namespace

extern "C" int vitae_ddp_available(void) { return bind() ? 1 : 0; }

extern "C" int vitae_ddp_unique_id(void* out128) {
    if (!out128) return VITAE_ERR_INVALID_ARG;
    if (!bind()) return VITAE_ERR_LAUNCH;
    UniqueId id;
    if (g.get_id(&id) != 0) return VITAE_ERR_LAUNCH;
    memcpy(out128, &id, sizeof(id));
    return VITAE_OK;
}

extern "C" int vitae_ddp_init(const void* unique_id128, int world_size, int rank) {
    if (!unique_id128 || world_size < 1 || rank < 0 || rank >= world_size) return VITAE_ERR_INVALID_ARG;
    if (!bind()) return VITAE_ERR_LAUNCH;
    if (g.comm) { g.destroy(g.comm); g.comm = nullptr; }
    UniqueId id;
    memcpy(&id, unique_id128, sizeof(id));
    if (g.init_rank(&g.comm, world_size, id, rank) != 0) { g.comm = nullptr; return VITAE_ERR_LAUNCH; }
    if (!g.events) {
        for (int i = 0; i < State::NEV; ++i)
            if (hipEventCreateWithFlags(&g.ev[i], hipEventDisableTiming) != hipSuccess) return VITAE_ERR_LAUNCH;
        g.events = true;
    }
    g.world = world_size; g.rank = rank;
    return VITAE_OK;
}

extern "C" int vitae_ddp_world_size(void) { return g.comm ? g.world : 0; }

extern "C" int vitae_ddp_allreduce_bucket(void* buf, long count, int is_bf16, void* compute_stream, void* comm_stream) {
    if (!buf || count <= 0 || !comm_stream) return VITAE_ERR_INVALID_ARG;
    if (!g.comm) return VITAE_ERR_INVALID_ARG;
    hipStream_t cs = (hipStream_t)compute_stream, ns = (hipStream_t)comm_stream;
    bool ring_ok;
    hipEvent_t e = next_event(cs, &ring_ok);
    if (!ring_ok) return VITAE_ERR_UNSUPPORTED_SHAPE;      // more fork / join points in one captured step than the event ring holds
    // fork: the bucket is final once everything enqueued on the compute stream so far has run
    if (hipEventRecord(e, cs) != hipSuccess || hipStreamWaitEvent(ns, e, 0) != hipSuccess) return VITAE_ERR_LAUNCH;
    if (g.all_reduce(buf, buf, (size_t)count, is_bf16 ? NCCL_BF16 : NCCL_F32, NCCL_SUM, g.comm, ns) != 0) return VITAE_ERR_LAUNCH;
    return VITAE_OK;
}

extern "C" int vitae_ddp_wait(void* compute_stream, void* comm_stream) {
    if (!comm_stream || !g.events) return VITAE_ERR_INVALID_ARG;
    bool ring_ok;
    hipEvent_t e = next_event((hipStream_t)compute_stream, &ring_ok);
    if (!ring_ok) return VITAE_ERR_UNSUPPORTED_SHAPE;
    // join: whatever the compute stream does next sees every bucket reduced so far
    if (hipEventRecord(e, (hipStream_t)comm_stream) != hipSuccess ||
        hipStreamWaitEvent((hipStream_t)compute_stream, e, 0) != hipSuccess)
        return VITAE_ERR_LAUNCH;
    return VITAE_OK;
}

extern "C" int vitae_ddp_destroy(void) {
    if (g.comm && g.destroy) g.destroy(g.comm);
    g.comm = nullptr; g.world = 0; g.rank = -1;
    return VITAE_OK;
}
