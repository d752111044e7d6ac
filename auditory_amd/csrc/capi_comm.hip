// C ABI, the reassembly collective of the sharded path: RCCL (bound lazily) and the direct device-to-device all-gather.
// See include/auditory_hip.h.
#include "capi_internal.h"

#include <cstdlib>

using namespace audc;

// ---- RCCL, bound lazily so that single-GPU users never load it ---------------------

namespace {
struct uid128 {
    char b[128];
};
typedef int (*rccl_get_uid_t)(uid128*);
typedef int (*rccl_comm_init_t)(void**, int, uid128, int);
typedef int (*rccl_comm_destroy_t)(void*);
typedef int (*rccl_allgather_t)(const void*, void*, size_t, int /*dtype*/, void*, hipStream_t);
typedef const char* (*rccl_errstr_t)(int);

void* rccl_open() {
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    return h;
}
constexpr int kNcclFloat32 = 7;  // ncclFloat32 in nccl.h / rccl.h

// ---- arrival flags of the direct all-gather ------------------------------------------------------------------------------
// Rank r's flag block: word [p * kFlagPitch] = the last step whose slab rank p has pushed into r's receive area (written by p,
// over xGMI, as a system-scope release behind its copy on the same stream); word [n_ranks * kFlagPitch] = r's own step
// counter; word [(n_ranks + 1) * kFlagPitch] = waits of r that ran into their poll bound.  One flag per 64-byte line.
constexpr int kFlagPitch = 16;
constexpr int kMaxGatherRanks = 64;

__global__ void k_gather_begin(unsigned* step) { *step += 1u; }

__global__ void k_gather_signal(unsigned* peer_flag, const unsigned* step) {
    __hip_atomic_store(peer_flag, *step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one lane per peer polls that peer's flag until it has reached this rank's step (every rank makes the same calls, so the
// counters run in lockstep); the poll count is bounded: a peer that never arrives is counted and the kernel ENDS
__global__ void k_gather_wait(const unsigned* flags, const unsigned* step, unsigned* timeouts, int n_ranks, int rank, long long max_polls) {
    const int p = int(threadIdx.x);
    if (p >= n_ranks || p == rank) return;
    const unsigned want = *step;
    for (long long i = 0; i < max_polls; ++i) {
        const unsigned v = __hip_atomic_load(flags + size_t(p) * kFlagPitch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (int(v - want) >= 0) return;
        __builtin_amdgcn_s_sleep(32);
    }
    atomicAdd(timeouts, 1u);
}
}  // namespace


extern "C" {

int aud_comm_unique_id(char id[128]) {
    if (!id) return AUD_EINVAL;
    void* h = rccl_open();
    if (!h) return AUD_ERCCL;
    auto get = reinterpret_cast<rccl_get_uid_t>(dlsym(h, "ncclGetUniqueId"));
    if (!get) return AUD_ERCCL;
    uid128 u;
    std::memset(&u, 0, sizeof(u));
    if (get(&u) != 0) return AUD_ERCCL;
    std::memcpy(id, u.b, 128);
    return AUD_OK;
}

int aud_comm_init(aud_ctx* c, int n_ranks, int rank, const char id[128]) {
    if (!c || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return AUD_EINVAL;
    if (c->comm) return fail(c, AUD_EINVAL, "communicator already initialised");
    AUD_HIP(c, make_current(c));
    if (!c->rccl_lib) c->rccl_lib = rccl_open();
    if (!c->rccl_lib) return fail(c, AUD_ERCCL, "cannot load librccl.so");
    auto init = reinterpret_cast<rccl_comm_init_t>(dlsym(c->rccl_lib, "ncclCommInitRank"));
    if (!init) return fail(c, AUD_ERCCL, "ncclCommInitRank not found");
    uid128 u;
    std::memcpy(u.b, id, 128);
    void* comm = nullptr;
    const int r = init(&comm, n_ranks, u, rank);
    if (r != 0) {
        auto es = reinterpret_cast<rccl_errstr_t>(dlsym(c->rccl_lib, "ncclGetErrorString"));
        return fail(c, AUD_ERCCL, std::string("ncclCommInitRank: ") + (es ? es(r) : "error"));
    }
    c->comm = comm;
    c->n_ranks = n_ranks;
    c->rank = rank;
    return AUD_OK;
}

int aud_comm_destroy(aud_ctx* c) {
    if (!c) return AUD_EINVAL;
    if (c->comm && c->rccl_lib) {
        auto destroy = reinterpret_cast<rccl_comm_destroy_t>(dlsym(c->rccl_lib, "ncclCommDestroy"));
        if (destroy) destroy(c->comm);
    }
    c->comm = nullptr;
    return AUD_OK;
}

int aud_gather_create(aud_ctx* c, int n_ranks, int rank, int64_t slab_floats, float** recv, char handle[128]) {
    if (!c || !recv || !handle || n_ranks < 1 || rank < 0 || rank >= n_ranks || slab_floats < 1) return AUD_EINVAL;
    if (n_ranks > kMaxGatherRanks) return fail(c, AUD_EINVAL, "direct gather: at most 64 ranks");
    if (c->gather.recv) return fail(c, AUD_EINVAL, "gather buffer already created");
    AUD_HIP(c, make_current(c));
    aud_ctx::Gather& g = c->gather;
    AUD_HIP(c, hipMalloc(reinterpret_cast<void**>(&g.recv), size_t(2) * size_t(n_ranks) * size_t(slab_floats) * sizeof(float)));
    // the flags are polled by this GPU while peers store into them over xGMI: fine-grained (coherent) memory where the
    // runtime has it for exportable allocations, plain device memory otherwise (the accesses are system-scope atomics)
    const size_t flag_bytes = size_t(n_ranks + 2) * kFlagPitch * sizeof(unsigned);
    g.flags_fine = hipExtMallocWithFlags(reinterpret_cast<void**>(&g.flags), flag_bytes, hipDeviceMallocFinegrained) == hipSuccess;
    if (!g.flags_fine) {
        (void)hipGetLastError();
        g.flags = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&g.flags), flag_bytes) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(g.recv);
            g.recv = nullptr;
            return fail(c, AUD_ENOMEM, "direct gather: flag block");
        }
    }
    hipIpcMemHandle_t h[2];
    static_assert(sizeof(h[0]) == 64, "hipIpcMemHandle_t is 64 bytes");
    hipError_t e = hipMemset(g.flags, 0, flag_bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h[0], g.recv);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h[1], g.flags);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(g.recv);
        (void)hipFree(g.flags);
        g.recv = nullptr;
        g.flags = nullptr;
        return fail(c, AUD_EHIP, "hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
    }
    std::memcpy(handle, h, 128);
    g.n_ranks = n_ranks;
    g.rank = rank;
    g.slab = slab_floats;
    g.calls = 0;
    long long ms = 2000;
    if (const char* env = getenv("AUD_GATHER_WAIT_MS")) ms = std::max(1LL, atoll(env));
    g.max_polls = ms * 400;  // a poll = one system-scope load + s_sleep: 2-3 us
    g.peer.assign(size_t(n_ranks), nullptr);
    g.peer[size_t(rank)] = g.recv;
    g.peer_flags.assign(size_t(n_ranks), nullptr);
    g.peer_flags[size_t(rank)] = g.flags;
    g.streams.assign(size_t(n_ranks), nullptr);
    g.done.assign(size_t(n_ranks), nullptr);
    // all or nothing: a failure behind the allocation takes everything down again (aud_gather_destroy walks the
    // half-built state: null streams / events are skipped), so a later create does not find "already created"
    e = hipEventCreateWithFlags(&g.fork, hipEventDisableTiming);
    for (int p = 0; p < n_ranks && e == hipSuccess; ++p) {
        if (p == rank) continue;
        e = hipStreamCreateWithFlags(&g.streams[size_t(p)], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g.done[size_t(p)], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g.peer[size_t(rank)] = nullptr;  // (not an opened handle)
        g.peer_flags[size_t(rank)] = nullptr;
        (void)aud_gather_destroy(c);
        return fail(c, AUD_EHIP, std::string("aud_gather_create: ") + hipGetErrorString(e));
    }
    *recv = g.recv;
    return AUD_OK;
}

int aud_gather_open_peer(aud_ctx* c, int peer, const char handle[128]) {
    if (!c || !handle) return AUD_EINVAL;
    aud_ctx::Gather& g = c->gather;
    if (!g.recv) return fail(c, AUD_EINVAL, "aud_gather_create has not been called");
    if (peer < 0 || peer >= g.n_ranks || peer == g.rank) return fail(c, AUD_EINVAL, "peer must be another rank of the gather");
    if (g.peer[size_t(peer)]) return fail(c, AUD_EINVAL, "peer already opened");
    AUD_HIP(c, make_current(c));
    hipIpcMemHandle_t h[2];
    std::memcpy(h, handle, 128);
    void *p = nullptr, *f = nullptr;
    AUD_HIP(c, hipIpcOpenMemHandle(&p, h[0], hipIpcMemLazyEnablePeerAccess));
    if (hipIpcOpenMemHandle(&f, h[1], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipIpcCloseMemHandle(p);
        return fail(c, AUD_EHIP, "hipIpcOpenMemHandle (the peer's flag block)");
    }
    g.peer[size_t(peer)] = static_cast<float*>(p);
    g.peer_flags[size_t(peer)] = static_cast<unsigned*>(f);
    return AUD_OK;
}

int aud_allgather_direct_dev(aud_ctx* c, const float* send, int64_t count, int* slab_index, void* stream) {
    if (!c || count < 0) return AUD_EINVAL;
    aud_ctx::Gather& g = c->gather;
    if (!g.recv) return fail(c, AUD_EINVAL, "aud_gather_create has not been called");
    if (count > g.slab) return fail(c, AUD_EINVAL, "count exceeds the slab the gather was created for");
    for (int p = 0; p < g.n_ranks; ++p)
        if (!g.peer[size_t(p)]) return fail(c, AUD_EINVAL, "a peer's receive buffer has not been opened");
    if (count > 0 && !send) return fail(c, AUD_EINVAL, "null buffer");
    AUD_HIP(c, make_current(c));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int which = int(g.calls & 1u);
    g.calls += 1;
    if (slab_index) *slab_index = which;
    const size_t bytes = size_t(count) * sizeof(float);
    const size_t slot = (size_t(which) * size_t(g.n_ranks) + size_t(g.rank)) * size_t(g.slab);
    unsigned* step = g.flags + size_t(g.n_ranks) * kFlagPitch;
    // this rank's step number advances ON THE DEVICE (a captured call replays with the next numbers); own slot on the caller's
    // stream; one push per peer, each on its own stream (its own xGMI link) with the arrival signal behind it, forked from and
    // joined back into the caller's stream
    hipLaunchKernelGGL(k_gather_begin, dim3(1), dim3(1), 0, st, step);
    AUD_HIP(c, hipGetLastError());
    if (bytes) AUD_HIP(c, hipMemcpyAsync(g.recv + slot, send, bytes, hipMemcpyDeviceToDevice, st));
    if (g.n_ranks > 1) AUD_HIP(c, hipEventRecord(g.fork, st));
    for (int p = 0; p < g.n_ranks; ++p) {
        if (p == g.rank) continue;
        hipStream_t sp = g.streams[size_t(p)];
        AUD_HIP(c, hipStreamWaitEvent(sp, g.fork, 0));
        if (bytes) AUD_HIP(c, hipMemcpyAsync(g.peer[size_t(p)] + slot, send, bytes, hipMemcpyDeviceToDevice, sp));
        hipLaunchKernelGGL(k_gather_signal, dim3(1), dim3(1), 0, sp, g.peer_flags[size_t(p)] + size_t(g.rank) * kFlagPitch, step);
        AUD_HIP(c, hipGetLastError());
        AUD_HIP(c, hipEventRecord(g.done[size_t(p)], sp));
        AUD_HIP(c, hipStreamWaitEvent(st, g.done[size_t(p)], 0));
    }
    return AUD_OK;
}

int aud_gather_wait_dev(aud_ctx* c, void* stream) {
    if (!c) return AUD_EINVAL;
    aud_ctx::Gather& g = c->gather;
    if (!g.recv) return fail(c, AUD_EINVAL, "aud_gather_create has not been called");
    if (g.n_ranks == 1) return AUD_OK;
    AUD_HIP(c, make_current(c));
    unsigned* step = g.flags + size_t(g.n_ranks) * kFlagPitch;
    hipLaunchKernelGGL(k_gather_wait, dim3(1), dim3(kMaxGatherRanks), 0, static_cast<hipStream_t>(stream), g.flags, step,
                       step + kFlagPitch, g.n_ranks, g.rank, g.max_polls);
    AUD_HIP(c, hipGetLastError());
    return AUD_OK;
}

int aud_gather_timeouts(aud_ctx* c, int* n) {
    if (!c || !n) return AUD_EINVAL;
    aud_ctx::Gather& g = c->gather;
    if (!g.recv) return fail(c, AUD_EINVAL, "aud_gather_create has not been called");
    AUD_HIP(c, make_current(c));
    AUD_HIP(c, hipDeviceSynchronize());
    unsigned v = 0;
    AUD_HIP(c, hipMemcpy(&v, g.flags + size_t(g.n_ranks + 1) * kFlagPitch, sizeof(v), hipMemcpyDeviceToHost));
    *n = int(v);
    return AUD_OK;
}

int aud_gather_destroy(aud_ctx* c) {
    if (!c) return AUD_EINVAL;
    aud_ctx::Gather& g = c->gather;
    if (!g.recv) return AUD_OK;
    (void)hipSetDevice(c->device);
    for (int p = 0; p < g.n_ranks; ++p) {
        if (p == g.rank) continue;
        if (g.streams[size_t(p)]) {
            (void)hipStreamSynchronize(g.streams[size_t(p)]);
            (void)hipStreamDestroy(g.streams[size_t(p)]);
        }
        if (g.done[size_t(p)]) (void)hipEventDestroy(g.done[size_t(p)]);
        if (g.peer[size_t(p)]) (void)hipIpcCloseMemHandle(g.peer[size_t(p)]);
        if (g.peer_flags[size_t(p)]) (void)hipIpcCloseMemHandle(g.peer_flags[size_t(p)]);
    }
    if (g.fork) (void)hipEventDestroy(g.fork);
    (void)hipFree(g.recv);
    if (g.flags) (void)hipFree(g.flags);
    g = aud_ctx::Gather();
    return AUD_OK;
}

int aud_allgather_dev(aud_ctx* c, const float* send, float* recv, int64_t count, void* stream) {
    if (!c || count < 0) return AUD_EINVAL;
    if (!c->comm) return fail(c, AUD_ERCCL, "aud_comm_init has not been called");
    if (count == 0) return AUD_OK;
    if (!send || !recv) return fail(c, AUD_EINVAL, "null buffer");
    AUD_HIP(c, make_current(c));
    auto ag = reinterpret_cast<rccl_allgather_t>(dlsym(c->rccl_lib, "ncclAllGather"));
    if (!ag) return fail(c, AUD_ERCCL, "ncclAllGather not found");
    const int r = ag(send, recv, size_t(count), kNcclFloat32, c->comm, static_cast<hipStream_t>(stream));
    if (r != 0) {
        auto es = reinterpret_cast<rccl_errstr_t>(dlsym(c->rccl_lib, "ncclGetErrorString"));
        return fail(c, AUD_ERCCL, std::string("ncclAllGather: ") + (es ? es(r) : "error"));
    }
    return AUD_OK;
}

}  // extern "C"

