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

// One lane per peer polls that peer's flag until it has reached this rank's step (every rank makes the same calls, so the
// counters run in lockstep).  The poll count is bounded: a peer that never arrives is counted and the kernel ENDS -- but not
// quietly: the late peers' slots of the step's slab are filled with NaN (whatever is queued behind this wait then computes NaN
// instead of plausible numbers from a stale slab), and a status word in host-mapped memory tells the next library call, which
// refuses to go on (AUD_EBROKEN) without having to synchronise anything.
__global__ void k_gather_wait(const unsigned* flags, const unsigned* step, unsigned* timeouts, volatile unsigned* host_status,
                              float* recv, long long slab, int n_ranks, int rank, long long max_polls) {
    int* late = reinterpret_cast<int*>(aud::dyn_lds());  // [kMaxGatherRanks] (dynamic LDS: the launch gives it)
    const int p = int(threadIdx.x);
    const unsigned want = *step;
    const bool mine = p < n_ranks && p != rank;
    bool arrived = !mine;
    for (long long i = 0; mine && i < max_polls; ++i) {
        const unsigned v = __hip_atomic_load(flags + size_t(p) * kFlagPitch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (int(v - want) >= 0) {
            arrived = true;
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
    late[p] = arrived ? 0 : 1;
    if (!arrived) atomicAdd(timeouts, 1u);
    __syncthreads();
    unsigned mask_lo = 0, mask_hi = 0;
    for (int q = 0; q < n_ranks; ++q)
        if (late[q]) (q < 32 ? mask_lo : mask_hi) |= 1u << (q & 31);
    if ((mask_lo | mask_hi) == 0) return;  // the common case: every peer's slab of this step is here
    const int which = int((want - 1u) & 1u);  // the slab this step uses (aud_allgather_direct_dev: calls & 1 before the increment)
    const float poison = __builtin_nanf("");
    for (int q = 0; q < n_ranks; ++q) {
        if (!late[q]) continue;
        float* slot = recv + (size_t(which) * size_t(n_ranks) + size_t(q)) * size_t(slab);
        for (long long i = p; i < slab; i += kMaxGatherRanks) slot[i] = poison;
    }
    if (p == 0) {
        host_status[1] = mask_lo;
        host_status[2] = mask_hi;
        __threadfence_system();
        host_status[0] = want ? want : 1u;  // non-zero: sticky
        __threadfence_system();
    }
}

// the gather's sticky failure state, checked (and refreshed from the host-mapped status word) at the top of every per-step call
int gather_usable(aud_ctx* c) {
    aud_ctx::Gather& g = c->gather;
    if (!g.broken && g.host_status && g.host_status[0] != 0) {
        g.broken = true;
        char buf[160];
        std::snprintf(buf, sizeof(buf), "direct gather: the wait of step %u ran into its poll bound (late peers: mask 0x%08x%08x); their "
                                        "slots were filled with NaN", g.host_status[0], g.host_status[2], g.host_status[1]);
        g.broken_why = buf;
    }
    if (g.broken) return fail(c, AUD_EBROKEN, g.broken_why + " -- aud_gather_destroy / aud_gather_create to go on");
    return AUD_OK;
}

int gather_break(aud_ctx* c, const std::string& why) {
    c->gather.broken = true;
    c->gather.broken_why = why;
    return fail(c, AUD_EBROKEN, why);
}

// calls of the current stream capture: an odd number in a finished capture breaks the slab alternation across its replays
int gather_note_capture(aud_ctx* c, hipStream_t st) {
    aud_ctx::Gather& g = c->gather;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(st, &cs, &id) != hipSuccess) {
        (void)hipGetLastError();
        cs = hipStreamCaptureStatusNone;
    }
    const bool capturing = cs == hipStreamCaptureStatusActive;
    if (g.cap_id != 0 && (!capturing || id != g.cap_id)) {  // the previous capture sequence is over
        const unsigned n = g.cap_calls;
        g.cap_id = 0;
        g.cap_calls = 0;
        if (n & 1u) return gather_break(c, "direct gather: a captured sequence held an ODD number of steps (" + std::to_string(n) +
                                               "): its replays would not alternate between the two receive slabs");
    }
    if (capturing) {
        g.cap_id = id;
        g.cap_calls += 1;
    }
    return AUD_OK;
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
    // the flags are polled by this GPU while peers store into them over xGMI: fine-grained (coherent) memory.  Ordinary device
    // memory is cached in this GPU's L2, where a peer's store is not guaranteed to become visible to the poll: only on request
    const size_t flag_bytes = size_t(n_ranks + 2) * kFlagPitch * sizeof(unsigned);
    g.flags_fine = hipExtMallocWithFlags(reinterpret_cast<void**>(&g.flags), flag_bytes, hipDeviceMallocFinegrained) == hipSuccess;
    if (!g.flags_fine) {
        (void)hipGetLastError();
        g.flags = nullptr;
        const char* coarse = getenv("AUD_GATHER_COARSE_FLAGS");
        if (!coarse || atoi(coarse) != 1) {
            (void)hipFree(g.recv);
            g.recv = nullptr;
            return fail(c, AUD_EHIP, "direct gather: no fine-grained device memory for the arrival flags (hipExtMallocWithFlags "
                                     "hipDeviceMallocFinegrained failed); AUD_GATHER_COARSE_FLAGS=1 takes ordinary device memory");
        }
        if (hipMalloc(reinterpret_cast<void**>(&g.flags), flag_bytes) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(g.recv);
            g.recv = nullptr;
            return fail(c, AUD_ENOMEM, "direct gather: flag block");
        }
    }
    hipIpcMemHandle_t h[2];
    static_assert(sizeof(h[0]) == 64, "hipIpcMemHandle_t is 64 bytes");
    const char* what = "hipMemset (flag block)";
    hipError_t e = hipMemset(g.flags, 0, flag_bytes);
    if (e == hipSuccess) { what = "hipDeviceSynchronize"; e = hipDeviceSynchronize(); }
    if (e == hipSuccess) { what = "hipIpcGetMemHandle (receive area; HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"; e = hipIpcGetMemHandle(&h[0], g.recv); }
    if (e == hipSuccess) { what = "hipIpcGetMemHandle (flag block; HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"; e = hipIpcGetMemHandle(&h[1], g.flags); }
    void* hst = nullptr;
    if (e == hipSuccess) { what = "hipHostMalloc (status word)"; e = hipHostMalloc(&hst, 64, hipHostMallocDefault); }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(g.recv);
        (void)hipFree(g.flags);
        g.recv = nullptr;
        g.flags = nullptr;
        return fail(c, AUD_EHIP, std::string("aud_gather_create: ") + what + ": " + hipGetErrorString(e));
    }
    std::memset(hst, 0, 64);
    g.host_status = static_cast<volatile unsigned*>(hst);
    g.broken = false;
    g.broken_why.clear();
    g.cap_id = 0;
    g.cap_calls = 0;
    std::memcpy(handle, h, 128);
    g.n_ranks = n_ranks;
    g.rank = rank;
    g.slab = slab_floats;
    g.calls = 0;
    long long ms = 30000;  // generous: a peer may be loading data, page-faulting or still capturing its graph; the variable
    if (const char* env = getenv("AUD_GATHER_WAIT_MS")) ms = std::max(1LL, atoll(env));  // is there to SHORTEN it (tests) or lengthen it
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
    int rc = gather_usable(c);
    if (rc != AUD_OK) return rc;
    AUD_HIP(c, make_current(c));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = gather_note_capture(c, st)) != AUD_OK) return rc;
    const int which = int(g.calls & 1u);
    if (slab_index) *slab_index = which;
    const size_t bytes = size_t(count) * sizeof(float);
    const size_t slot = (size_t(which) * size_t(g.n_ranks) + size_t(g.rank)) * size_t(g.slab);
    unsigned* step = g.flags + size_t(g.n_ranks) * kFlagPitch;
    // this rank's step number advances ON THE DEVICE (a captured call replays with the next numbers); own slot on the caller's
    // stream; one push per peer, each on its own stream (its own xGMI link) with the arrival signal behind it, forked from and
    // joined back into the caller's stream.  From the first enqueue on a failure leaves the device-side step count (and
    // possibly some pushes) ahead of the host's and of the peers': the gather is then BROKEN, not quietly out of step.
    hipError_t e = hipSuccess;
    const char* what = "k_gather_begin";
    hipLaunchKernelGGL(k_gather_begin, dim3(1), dim3(1), 0, st, step);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(c, e, "aud_allgather_direct_dev: k_gather_begin");  // (nothing was queued: still usable)
    auto step_ok = [&](hipError_t r, const char* w) {
        if (e == hipSuccess && r != hipSuccess) {
            e = r;
            what = w;
        }
        return e == hipSuccess;
    };
    if (bytes) step_ok(hipMemcpyAsync(g.recv + slot, send, bytes, hipMemcpyDeviceToDevice, st), "copy into the own slot");
    if (g.n_ranks > 1 && e == hipSuccess) step_ok(hipEventRecord(g.fork, st), "hipEventRecord (fork)");
    for (int p = 0; p < g.n_ranks && e == hipSuccess; ++p) {
        if (p == g.rank) continue;
        hipStream_t sp = g.streams[size_t(p)];
        if (!step_ok(hipStreamWaitEvent(sp, g.fork, 0), "hipStreamWaitEvent (fork)")) break;
        if (bytes && !step_ok(hipMemcpyAsync(g.peer[size_t(p)] + slot, send, bytes, hipMemcpyDeviceToDevice, sp), "push to a peer")) break;
        hipLaunchKernelGGL(k_gather_signal, dim3(1), dim3(1), 0, sp, g.peer_flags[size_t(p)] + size_t(g.rank) * kFlagPitch, step);
        if (!step_ok(hipGetLastError(), "k_gather_signal")) break;
        if (!step_ok(hipEventRecord(g.done[size_t(p)], sp), "hipEventRecord (join)")) break;
        step_ok(hipStreamWaitEvent(st, g.done[size_t(p)], 0), "hipStreamWaitEvent (join)");
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return gather_break(c, std::string("direct gather: ") + what + " failed behind the step's first enqueue (" + hipGetErrorString(e) +
                                   "): this rank's step count no longer matches its peers'");
    }
    g.calls += 1;  // (only a step that was queued in full counts)
    return AUD_OK;
}

int aud_gather_wait_dev(aud_ctx* c, void* stream) {
    if (!c) return AUD_EINVAL;
    aud_ctx::Gather& g = c->gather;
    if (!g.recv) return fail(c, AUD_EINVAL, "aud_gather_create has not been called");
    int rc = gather_usable(c);
    if (rc != AUD_OK) return rc;
    if (g.n_ranks == 1) return AUD_OK;
    AUD_HIP(c, make_current(c));
    unsigned* step = g.flags + size_t(g.n_ranks) * kFlagPitch;
    hipLaunchKernelGGL(k_gather_wait, dim3(1), dim3(kMaxGatherRanks), kMaxGatherRanks * sizeof(int), static_cast<hipStream_t>(stream), g.flags, step,
                       step + kFlagPitch, g.host_status, g.recv, static_cast<long long>(g.slab), g.n_ranks, g.rank, g.max_polls);
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

int aud_gather_flags_fine(aud_ctx* c) {
    if (!c || !c->gather.recv) return -1;
    return c->gather.flags_fine ? 1 : 0;
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
    if (g.host_status) (void)hipHostFree(const_cast<unsigned*>(g.host_status));
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

