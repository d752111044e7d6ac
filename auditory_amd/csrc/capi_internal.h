// Shared by the translation units of the C ABI (capi.hip: contexts, plans, device entry points; capi_host.hip: the
// host-staged entry points; capi_comm.hip: RCCL and the direct all-gather; wave_tables.hip: the wave kernels' tables).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <new>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

struct aud_ctx {
    int device = -1;
    std::string err = "";          // last error message, under err_mutex (entry points may fail concurrently)
    std::mutex err_mutex;
    hipStream_t stream = nullptr;  // used by the _host entry points
    std::mutex host_mutex;         // ... which serialise on it (HostCallGuard)
    // grow-only device workspaces for the _host entry points
    void* ws[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t ws_cap[4] = {0, 0, 0, 0};
    // pinned host staging for the _host entry points' result copies (grow-only): device-to-host copies into pinned memory
    // run at the link's rate and overlap with the widening of the previous chunk
    void* pin = nullptr;
    size_t pin_cap = 0;
    hipEvent_t pin_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // aud_host_alloc / aud_host_register: pinned, device-visible host memory the caller keeps result tensors in (under
    // host_mutex).  `dev` is the address the device uses for `p` (the same for aud_host_alloc memory; what
    // hipHostGetDevicePointer says for a registered range, e.g. a shared mapping several processes write their shards into)
    struct HostBlock {
        unsigned char* p;
        size_t bytes;
        unsigned char* dev;
        bool registered;
    };
    std::vector<HostBlock> host_blocks;
    // RCCL (loaded lazily)
    void* rccl_lib = nullptr;
    void* comm = nullptr;
    int n_ranks = 0, rank = 0;
    // direct all-gather (aud_gather_*): this rank's receive buffer, the peers' mapped ones, one stream + event per peer
    struct Gather {
        int n_ranks = 0, rank = 0;
        int64_t slab = 0;
        float* recv = nullptr;             // [2][n_ranks][slab]: consecutive steps alternate between the two slabs
        unsigned* flags = nullptr;         // kFlagPitch words per rank: [p] = the last step rank p pushed here; behind them
                                           // this rank's own step counter and its count of timed-out waits
        bool flags_fine = false;           // the flag block is fine-grained memory (else plain hipMalloc)
        std::vector<float*> peer;          // [n_ranks], peer[rank] = recv
        std::vector<unsigned*> peer_flags; // [n_ranks], peer_flags[rank] = flags
        std::vector<hipStream_t> streams;  // [n_ranks], null at `rank`
        std::vector<hipEvent_t> done;      // [n_ranks]
        hipEvent_t fork = nullptr;
        unsigned calls = 0;                // host side: steps issued (slab = calls & 1)
        long long max_polls = 0;
        // sticky failure state: host_status is host-mapped memory the wait kernel writes on a time-out ([0] = the step that
        // timed out, [1] / [2] = the late peers' bit mask), read by the next call without synchronising; `broken` once seen,
        // or after a call that failed behind its first enqueue
        volatile unsigned* host_status = nullptr;
        bool broken = false;
        std::string broken_why;
        // calls captured into the current capture sequence (its id from hipStreamGetCaptureInfo): an odd count breaks the
        // slab alternation across replays
        unsigned long long cap_id = 0;
        unsigned cap_calls = 0;
    } gather;
};

// a signal resident on the device between calls (aud_signal_upload).  The context keeps a list of its live signals:
// aud_shutdown frees their device memory and detaches them (ctx = nullptr), so that an aud_signal_destroy BEHIND the shutdown
// -- a destructor running late -- frees the handle and touches nothing of the dead context
struct aud_signal {
    aud_ctx* ctx = nullptr;
    void* d = nullptr;
    int dtype = 0;
    int64_t n = 0;
    // aud_signal_sync: capacity of `d`, and the host SHADOW of what `d` holds (malloc; valid only while shadow_ok: an upload
    // that failed half-way leaves the device copy undefined, and the next sync uploads everything)
    size_t cap = 0;
    unsigned char* shadow = nullptr;
    size_t shadow_cap = 0;
    bool shadow_ok = false;
    ~aud_signal() { std::free(shadow); }
};
// every handle that is alive and the context it belongs to (a destroy on a detached handle must not touch a dead context's mutex)
struct SignalRegistry {
    std::mutex m;
    std::vector<aud_signal*> live;
    static SignalRegistry& get() {
        static SignalRegistry r;
        return r;
    }
};

struct aud_plan {
    aud_ctx* ctx = nullptr;
    aud_plan_desc d{};
    int H = 0, M = 0, ratio = 0;
    int nfac = 0;
    int fac[aud::kMaxFactors] = {0};
    int F_generic = 0;         // frames a workgroup of the any-N kernel takes on the route the plan runs (generic_route())
    // smooth window lengths in place (kernels.h MelspecArgs::ip_*): F_ip = 0 where the route does not serve the length
    int F_two = 0, F_ip = 0, ip_nfac = 0;
    int ip_fac[aud::kMaxFactors] = {0};
    bool direct = false;       // no LDS-resident transform fits this window length: the O(N H) kernel (melspec_direct.hip)
    void* d_tw64 = nullptr;    // its [N] complex<double> table
    int ip_opt = 1;            // plan option "plain_inplace": 1 (default) the in-place route where it serves, 0 the two-buffer route
    // generic kernel, Bluestein route (kernels.h MelspecArgs::bl_*): 0 = not used
    int bl_L = 0, bl_nfac = 0;
    bool bl_inplace = false;
    int bl_fac[aud::kMaxFactors] = {0};
    void* d_bl_chirp = nullptr;
    void* d_bl_bhat = nullptr;
    void* d_bl_tw = nullptr;
    void* d_bl_fix = nullptr;  // tables of the fixed-geometry chirp kernel (melspec_chirp.hip) where it serves the plan
    void* d_fix_chirp = nullptr;  // ... and its chirp of length N (the any-N route's, where it has one, is of length M)
    int F_bl = 0;              // frames per workgroup of the any-N kernel's Bluestein route
    int chirp_opt = 1;         // plan option "chirp_kernel": 1 (default) use it where it serves, 0 the any-N route
    int xcd_remap = 1;         // workgroup -> tile order keeps an XCD on one run of tiles (kernels.h)
    // wave-autonomous kernel of this window length (melspec_wave.hip), the default where it exists
    int wave_kind = 0;         // = the kind number of melspec_wave.hip (0: none)
    bool use_wave = false;     // false: the generic kernel (no wave kernel, or plan option "kernel" = 1)
    aud::WaveArgs wv{};
    void* d_blob = nullptr;   // wave kernels: every read-only table, laid out like its LDS copy (kernels.h WaveArgs)
    void* d_gtab = nullptr;   // w64x16: lane-ordered pass-1 and split twiddles read from global memory
    void* d_tw = nullptr;
    void* d_filt = nullptr;
    int32_t* d_bin_pts = nullptr;
    void* d_gabor = nullptr;
    float* d_gabor32 = nullptr;  // float32 copy of the taps (the fused gabor phase of the item kernel, gabor_tile.h)
    // workgroup-per-item variant of the N = 400 kernel (kernels.h ItemArgs): launch shape, whether it exists for this plan,
    // and plan option "item_kernel": -1 / 0 = not used (the default: measured slower than the tile kernel at 256 items per
    // launch), 1 = wherever it exists: aud_process_batch_dev becomes ONE launch with Convolve fused in, mel-only calls too
    aud::ItemArgs itm{};
    bool has_item = false;
    int item_opt = -1;
    int lds_pad = 0;     // plan option "lds_pad": extra dynamic LDS per workgroup of the wave kernels (occupancy experiments)
    int gabor_opt = -1;  // plan option "gabor_kernel" (kernels.h GaborArgs::mode): -1 = by compute type
    int fused_tail_opt = -1;  // plan option "fused_tail": 0 = aud_segment_batch_dev always runs the tail on the stored tensors
    void* d_dct = nullptr;  // [mfcc_coefs][nf] DCT-I rows
    unsigned long long stamps = 0;  // diagnostic builds (-DAUD_STAMPS): device buffer for the phase stamps
    const char* family = "generic";
};

namespace audc {

// buffer element holding the last sample of an item's stream (-1 for an empty stream)
inline int64_t item_last(const aud_item& it) {
    if (it.sig_len <= 0) return it.sig_off - 1;
    return it.sig_off + int64_t(it.sig_len - 1) * (it.sig_stride > 1 ? it.sig_stride : 1);
}

inline int fail(aud_ctx* c, int code, const std::string& msg) {
    if (c) {
        std::lock_guard<std::mutex> lk(c->err_mutex);
        c->err = msg;
    }
    return code;
}

inline int hip_fail(aud_ctx* c, hipError_t e, const char* what) {
    return fail(c, AUD_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

// Host entry points queue asynchronous copies from / to caller-owned (pageable) memory on the context's stream: on EVERY
// exit behind the first such copy the stream is drained, so the caller may free or reuse its buffers whatever the status.
// They also serialise on the context: its workspaces and stream are shared state (two goroutines on one aud_ctx).
struct HostCallGuard {
    aud_ctx* c;
    explicit HostCallGuard(aud_ctx* ctx) : c(ctx) { c->host_mutex.lock(); }
    ~HostCallGuard() {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        c->host_mutex.unlock();
    }
};

#define AUD_HIP(c, call)                                   \
    do {                                                   \
        hipError_t e__ = (call);                           \
        if (e__ != hipSuccess) return hip_fail(c, e__, #call); \
    } while (0)

// Goroutines / Python threads migrate between OS threads, so every entry point makes its device
// current -- but only when it is not already (the device entry points may run under stream
// capture, where needless runtime calls are best avoided).
inline hipError_t make_current(const aud_ctx* c) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == c->device) return hipSuccess;
    return hipSetDevice(c->device);
}

inline int ensure_ws(aud_ctx* c, int slot, size_t bytes) {
    if (c->ws_cap[slot] >= bytes) return AUD_OK;
    if (c->ws[slot]) {
        AUD_HIP(c, hipFree(c->ws[slot]));
        c->ws[slot] = nullptr;
        c->ws_cap[slot] = 0;
    }
    const size_t cap = bytes + bytes / 4 + 4096;
    AUD_HIP(c, hipMalloc(&c->ws[slot], cap));
    c->ws_cap[slot] = cap;
    return AUD_OK;
}

inline int ensure_pin(aud_ctx* c, size_t bytes) {
    if (c->pin_cap >= bytes) return AUD_OK;
    if (c->pin) {
        AUD_HIP(c, hipHostFree(c->pin));
        c->pin = nullptr;
        c->pin_cap = 0;
    }
    // (the events first: a failure here must not leave pin_cap saying "ready" with null events behind it)
    for (auto& e : c->pin_ev)
        if (!e) AUD_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const size_t cap = bytes + bytes / 4 + 4096;
    AUD_HIP(c, hipHostMalloc(&c->pin, cap, hipHostMallocDefault));
    c->pin_cap = cap;
    return AUD_OK;
}

// float32 -> float64, the inner loop of every host entry point's result path (1 M values per 256 utterances): eight values
// per iteration where the CPU has AVX2 (checked once at run time; the library itself is built for baseline x86-64)
#if defined(__x86_64__) && !defined(AUD_EMUL_NO_AVX2)
#include <immintrin.h>
__attribute__((target("avx2"))) inline void widen_avx2(double* d, const float* s, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 v = _mm256_loadu_ps(s + i);
        _mm256_storeu_pd(d + i, _mm256_cvtps_pd(_mm256_castps256_ps128(v)));
        _mm256_storeu_pd(d + i + 4, _mm256_cvtps_pd(_mm256_extractf128_ps(v, 1)));
    }
    for (; i < n; ++i) d[i] = double(s[i]);
}
inline void widen(double* d, const float* s, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return widen_avx2(d, s, n);
    for (size_t i = 0; i < n; ++i) d[i] = double(s[i]);
}
#else
inline void widen(double* d, const float* s, size_t n) {
    for (size_t i = 0; i < n; ++i) d[i] = double(s[i]);
}
#endif

// Device float32 results -> the caller's float64 tensors: up to eight chunks copied into pinned staging back to back on the
// context's stream, each widened by this thread as soon as ITS copy has landed, while the later ones are still in flight.
// `parts`: destination (null: skipped), element count, in device order starting at d_src.
struct WidenPart {
    double* dst;
    size_t n;
};
inline int fetch_widened(aud_ctx* c, const float* d_src, const WidenPart* parts, int n_parts) {
    size_t total = 0;
    for (int i = 0; i < n_parts; ++i) total += parts[i].n;
    if (total == 0) return AUD_OK;
    int rc = ensure_pin(c, total * 4);
    if (rc != AUD_OK) return rc;
    float* h = static_cast<float*>(c->pin);
    constexpr int kMaxChunks = 8;
    const int chunks = total >= (size_t(1) << 18) ? kMaxChunks : 1;
    const size_t per = (total + chunks - 1) / chunks;
    for (int k = 0; k < chunks; ++k) {
        const size_t lo = size_t(k) * per, hi = std::min(total, lo + per);
        if (hi > lo) AUD_HIP(c, hipMemcpyAsync(h + lo, d_src + lo, (hi - lo) * 4, hipMemcpyDeviceToHost, c->stream));
        AUD_HIP(c, hipEventRecord(c->pin_ev[k], c->stream));
    }
    size_t part_lo = 0;
    int k_done = -1;
    for (int i = 0; i < n_parts; ++i) {
        const size_t part_hi = part_lo + parts[i].n;
        if (parts[i].dst) {
            size_t pos = part_lo;
            while (pos < part_hi) {
                const int k = int(pos / per);
                if (k > k_done) {
                    AUD_HIP(c, hipEventSynchronize(c->pin_ev[k]));
                    k_done = k;
                }
                const size_t end = std::min(part_hi, size_t(k + 1) * per);
                widen(parts[i].dst + (pos - part_lo), h + pos, end - pos);
                pos = end;
            }
        }
        part_lo = part_hi;
    }
    return AUD_OK;
}

// the aud_host_alloc / aud_host_register block of the context that holds [p, p + bytes), or null (the caller holds host_mutex)
inline const aud_ctx::HostBlock* find_host_block(const aud_ctx* c, const void* p, size_t bytes) {
    const unsigned char* q = static_cast<const unsigned char*>(p);
    for (const auto& b : c->host_blocks)
        if (q >= b.p && q + bytes <= b.p + b.bytes) return &b;
    return nullptr;
}
inline bool in_host_block(const aud_ctx* c, const void* p, size_t bytes) { return find_host_block(c, p, bytes) != nullptr; }
// every requested (non-null, non-empty) part lies in aud_host_alloc memory
inline bool all_parts_pinned(const aud_ctx* c, const WidenPart* parts, int n_parts) {
    bool any = false;
    for (int i = 0; i < n_parts; ++i) {
        if (!parts[i].dst || parts[i].n == 0) continue;
        if (!in_host_block(c, parts[i].dst, parts[i].n * sizeof(double))) return false;
        any = true;
    }
    return any;
}
// device float32 results -> float64 in the caller's PINNED tensors, written by the device (smooth_mel.hip launch_widen_to_host)
inline int store_widened(aud_ctx* c, const float* d_src, const WidenPart* parts, int n_parts) {
    size_t lo = 0;
    for (int i = 0; i < n_parts; ++i) {
        if (parts[i].dst && parts[i].n) {
            const aud_ctx::HostBlock* b = find_host_block(c, parts[i].dst, parts[i].n * sizeof(double));
            if (!b) return fail(c, AUD_EINVAL, "store_widened: a tensor left its pinned block");
            double* dst = reinterpret_cast<double*>(b->dev + (reinterpret_cast<unsigned char*>(parts[i].dst) - b->p));
            AUD_HIP(c, aud::launch_widen_to_host(d_src + lo, dst, parts[i].n, c->stream));
        }
        lo += parts[i].n;
    }
    AUD_HIP(c, hipStreamSynchronize(c->stream));
    return AUD_OK;
}

template <typename TT>
inline std::vector<TT> convert(const double* src, size_t n) {
    std::vector<TT> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = TT(src[i]);
    return v;
}

inline int upload(aud_ctx* c, void** dst, const void* src, size_t bytes) {
    AUD_HIP(c, hipMalloc(dst, bytes ? bytes : 16));
    if (bytes) AUD_HIP(c, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return AUD_OK;
}

inline int upload_real(aud_ctx* c, void** dst, const double* src, size_t n, int dt) {
    if (dt == AUD_F64) return upload(c, dst, src, n * 8);
    std::vector<float> v = convert<float>(src, n);
    return upload(c, dst, v.data(), n * 4);
}

// capi.hip
void fill_melspec_args(const aud_plan* p, aud::MelspecArgs* a);
hipError_t launch_frames(const aud_plan* p, const aud::MelspecArgs& a, hipStream_t st);
const char* plan_family(const aud_plan* p);
bool plain_inplace(const aud_plan* p);   // smooth window length on the any-N kernel's in-place route
void generic_route(aud_plan* p);         // F_generic of the route the plan's options select
// wave_tables.hip
int build_wave_tables(aud_plan* p, const int32_t* bin_pts, const double* mel_filters);

}  // namespace audc
