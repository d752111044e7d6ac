// C ABI of libauditory_hip.so: context / plan management and the batch entry points.
// See include/auditory_hip.h for the contract of every function.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
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
    // RCCL (loaded lazily)
    void* rccl_lib = nullptr;
    void* comm = nullptr;
    int n_ranks = 0, rank = 0;
    // direct all-gather (aud_gather_*): this rank's receive buffer, the peers' mapped ones, one stream + event per peer
    struct Gather {
        int n_ranks = 0, rank = 0;
        int64_t slab = 0;
        float* recv = nullptr;
        std::vector<float*> peer;          // [n_ranks], peer[rank] = recv
        std::vector<hipStream_t> streams;  // [n_ranks], null at `rank`
        std::vector<hipEvent_t> done;      // [n_ranks]
        hipEvent_t fork = nullptr;
    } gather;
};

struct aud_plan {
    aud_ctx* ctx = nullptr;
    aud_plan_desc d{};
    int H = 0, M = 0, ratio = 0;
    int nfac = 0;
    int fac[aud::kMaxFactors] = {0};
    int F_generic = 0;
    // generic kernel, Bluestein route (kernels.h MelspecArgs::bl_*): 0 = not used
    int bl_L = 0, bl_nfac = 0;
    int bl_fac[aud::kMaxFactors] = {0};
    void* d_bl_chirp = nullptr;
    void* d_bl_bhat = nullptr;
    void* d_bl_tw = nullptr;
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
    void* d_dct = nullptr;  // [mfcc_coefs][nf] DCT-I rows
    unsigned long long stamps = 0;  // diagnostic builds (-DAUD_STAMPS): device buffer for the phase stamps
    const char* family = "generic";
};

namespace {

// buffer element holding the last sample of an item's stream (-1 for an empty stream)
int64_t item_last(const aud_item& it) {
    if (it.sig_len <= 0) return it.sig_off - 1;
    return it.sig_off + int64_t(it.sig_len - 1) * (it.sig_stride > 1 ? it.sig_stride : 1);
}

int fail(aud_ctx* c, int code, const std::string& msg) {
    if (c) {
        std::lock_guard<std::mutex> lk(c->err_mutex);
        c->err = msg;
    }
    return code;
}

int hip_fail(aud_ctx* c, hipError_t e, const char* what) {
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
hipError_t make_current(const aud_ctx* c) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == c->device) return hipSuccess;
    return hipSetDevice(c->device);
}

int ensure_ws(aud_ctx* c, int slot, size_t bytes) {
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

template <typename TT>
std::vector<TT> convert(const double* src, size_t n) {
    std::vector<TT> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = TT(src[i]);
    return v;
}

int upload(aud_ctx* c, void** dst, const void* src, size_t bytes) {
    AUD_HIP(c, hipMalloc(dst, bytes ? bytes : 16));
    if (bytes) AUD_HIP(c, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return AUD_OK;
}

int upload_real(aud_ctx* c, void** dst, const double* src, size_t n, int dt) {
    if (dt == AUD_F64) return upload(c, dst, src, n * 8);
    std::vector<float> v = convert<float>(src, n);
    return upload(c, dst, v.data(), n * 4);
}

// as few Stockham stages as possible out of the radices the kernel has in registers
// (16, 8, 4, 2 | 25, 5 | 3), then whatever primes are left
void factorize(int m, int* fac, int* nfac) {
    int n = 0;
    while (m % 16 == 0) { fac[n++] = 16; m /= 16; }
    while (m % 8 == 0) { fac[n++] = 8; m /= 8; }
    while (m % 4 == 0) { fac[n++] = 4; m /= 4; }
    while (m % 2 == 0) { fac[n++] = 2; m /= 2; }
    while (m % 25 == 0) { fac[n++] = 25; m /= 25; }
    while (m % 5 == 0) { fac[n++] = 5; m /= 5; }
    while (m % 3 == 0) { fac[n++] = 3; m /= 3; }
    for (int p = 7; int64_t(p) * p <= m; p += 2)
        while (m % p == 0) { fac[n++] = p; m /= p; }
    if (m > 1) fac[n++] = m;
    *nfac = n;
}

void fill_melspec_args(const aud_plan* p, aud::MelspecArgs* a) {
    const aud_plan_desc& d = p->d;
    std::memset(a, 0, sizeof(*a));
    a->N = d.win_samples;
    a->S = d.step_samples;
    a->T = d.segment_steps;
    a->border = d.border_steps;
    a->H = p->H;
    a->M = p->M;
    a->ratio = p->ratio;
    a->nfac = p->nfac;
    for (int i = 0; i < p->nfac; ++i) a->fac[i] = p->fac[i];
    a->tw = p->d_tw;
    a->nf = d.mel.n_filters;
    a->bin_pts = p->d_bin_pts;
    a->filt = p->d_filt;
    a->mel_log_off = d.mel.log_off;
    a->mel_log_min = d.mel.log_min;
    a->renorm = d.mel.renorm;
    a->renorm_min = d.mel.renorm_min;
    a->renorm_scale = d.mel.renorm_scale;
    a->comp_log_pow = d.dft.comp_log_pow;
    a->dft_log_min = d.dft.log_min;
    a->dft_log_off = d.dft.log_offset;
    a->F = p->F_generic;
    a->bl_L = p->bl_L;
    a->bl_nfac = p->bl_nfac;
    for (int i = 0; i < p->bl_nfac; ++i) a->bl_fac[i] = p->bl_fac[i];
    a->bl_chirp = p->d_bl_chirp;
    a->bl_bhat = p->d_bl_bhat;
    a->bl_tw = p->d_bl_tw;
    a->xcd_remap = p->xcd_remap;
    a->stamps = reinterpret_cast<unsigned long long*>(p->stamps);
}

// the plan's frame -> power -> mel kernel (whatever family it selected), raw power, no smoothing
hipError_t launch_frames(const aud_plan* p, const aud::MelspecArgs& a, hipStream_t st) {
    if (p->use_wave && p->wave_kind) return aud::launch_melspec_wave(p->wave_kind, a, p->wv, p->d.compute_dtype, st);
    return aud::launch_melspec_generic(a, p->d.compute_dtype, st);
}

const char* plan_family(const aud_plan* p);

// Tables of the wave-autonomous kernels (melspec_wave.hip) as one blob that is copied verbatim into LDS:
//   w4     per filter group one row of FLOAT32 weights (both compute types; x 1/4, exact): the group's filters (its
//          slots) one after the other, each as aligned 4-bin chunks (zero weights outside [lo, hi]), slot k padded to
//          slot_steps[k] chunks in every group
//   slots  per group and slot: the filter's first P chunk and its id (w64x16: compact rows, see below)
//   twa    pass twiddles W_N^(2 j k1), [k1 - 1][j]; tws: split twiddles W_N^k, k <= N/4 (compute type)
// A plan whose tables do not fit (16-bit indices, LDS) simply has no wave kernel.
int build_wave_tables(aud_plan* p, const int32_t* bin_pts, const double* mel_filters) {
    aud_ctx* c = p->ctx;
    const aud_plan_desc& d = p->d;
    const int N = d.win_samples, nf = d.mel.n_filters, dt = d.compute_dtype;
    const int kind = aud::melspec_wave_kind(N);  // N = 512 -> w16x16; N = 400 -> w20x10; N = 2048 -> w64x16
    aud::WaveGeometry g;
    if (!kind || !aud::melspec_wave_geometry(kind, N, &g)) return AUD_OK;
    const size_t tsz = dt == AUD_F64 ? 8 : 4;  // twiddles
    const size_t wsz = 4;                       // weights: float32
    const int G = g.n_groups, p_chunks = (N / 2 + 1 + 3) / 4;  // chunks of a padded power row (kHp / 4 of the kernel)
    if (nf >= 0xFFFF || nf > 8 * G) return AUD_OK;  // more than eight filters per group: no wave kernel
    // chunks per filter; a filter without taps still takes one (all-zero) step: its sum is 0 + LogOff
    std::vector<int> c0(nf), nc(nf), order(nf);
    for (int f = 0; f < nf; ++f) {
        const int lo = bin_pts[f], hi = bin_pts[f + 2];
        c0[f] = hi >= lo ? lo >> 2 : 0;
        nc[f] = hi >= lo ? (hi >> 2) - (lo >> 2) + 1 : 1;
        if (c0[f] + nc[f] > p_chunks) return AUD_OK;  // table reaches past the spectrum: the generic path reports it
        order[f] = f;
    }
    // widest filters first, dealt round-robin: slot k of every group then holds filters of nearly equal width
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return nc[x] > nc[y]; });
    const int n_slots = std::max(1, (nf + G - 1) / G);
    aud::WaveArgs e{};
    int n_steps = 0;
    std::vector<int> slot_pos(n_slots);
    for (int k = 0; k < n_slots; ++k) {
        int mx = 1;
        for (int r = k * G; r < std::min(nf, (k + 1) * G); ++r) mx = std::max(mx, nc[order[r]]);
        if (mx > 255) return AUD_OK;
        e.slot_steps[k] = static_cast<unsigned char>(mx);
        slot_pos[k] = n_steps;
        n_steps += mx;
    }
    const bool compact = kind == 4;  // one filter group per lane: per-group rows of slot_steps chunks would be 2x the LDS
    size_t w_stride = 0;
    std::vector<double> wrows;
    std::vector<uint32_t> slots;
    if (!compact) {
        // weight rows: row stride an odd number of 16-byte pieces, so that the groups' reads of one step spread over the banks
        w_stride = size_t(n_steps) * 4 * wsz;
        if ((w_stride / 16) % 2 == 0) w_stride += 16;
        wrows.assign(size_t(G) * (w_stride / wsz), 0.0);
        slots.assign(size_t(G) * n_slots, 0xFFFFu << 16);
    } else {
        // compact rows: every filter keeps only its own chunks (from the first P chunk its slot reads), one shared all-zero
        // chunk serves the steps past a filter's end; slot record = {first P chunk | filter id << 16, row start in 16-byte
        // pieces | own chunks << 16}
        slots.assign(size_t(G) * n_slots * 2, 0u);
        for (int gi = 0; gi < G; ++gi)
            for (int k = 0; k < n_slots; ++k) slots[(size_t(gi) * n_slots + k) * 2] = 0xFFFFu << 16;
        wrows.assign(4, 0.0);  // the zero chunk, at piece 0
    }
    size_t wpieces = 4 * wsz / 16;        // compact rows: 16-byte pieces laid down so far
    unsigned used[8][4] = {};             // [slot][16-lane read group]: piece residues mod 16 taken
    for (int r = 0; r < nf; ++r) {
        const int f = order[r], k = r / G, gi = r % G, ns = e.slot_steps[k];
        const int lo = bin_pts[f], hi = bin_pts[f + 2];
        const int pc0 = std::min(c0[f], p_chunks - ns);  // every step of the slot reads inside the row
        double* wr;
        if (!compact) {
            slots[size_t(gi) * n_slots + k] = uint32_t(pc0) | (uint32_t(f) << 16);
            wr = &wrows[size_t(gi) * (w_stride / wsz) + size_t(slot_pos[k]) * 4];
        } else {
            const int own = hi >= lo ? c0[f] + nc[f] - pc0 : 0;  // chunks from pc0 to the filter's last one
            // a row may start on any 16-byte piece; its start is pushed forward (<= 15 pieces) until its piece index mod 16
            // differs from that of every earlier row of the same slot whose lane shares one of ds_read_b128's 16-lane
            // groups -- the lanes of a group then read 16 different bank quads at every step (modelled 11.7 -> 4 cycles)
            static const unsigned char kGroupOfLane[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0,
                                                           1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
            const int grp16 = 2 * (gi >> 5) + kGroupOfLane[gi & 31];
            size_t piece = wpieces;
            for (int tries = 0; tries < 16 && (used[k][grp16] >> (piece & 15) & 1u); ++tries) ++piece;
            used[k][grp16] |= 1u << (piece & 15);
            const size_t per_chunk = 4 * wsz / 16;  // 16-byte pieces per chunk
            if (piece + size_t(own) * per_chunk > 0xFFFF) return AUD_OK;
            slots[(size_t(gi) * n_slots + k) * 2] = uint32_t(pc0) | (uint32_t(f) << 16);
            slots[(size_t(gi) * n_slots + k) * 2 + 1] = uint32_t(piece) | (uint32_t(own) << 16);
            wpieces = piece + size_t(own) * per_chunk;
            wrows.resize(wpieces * 16 / wsz, 0.0);
            wr = wrows.data() + piece * 16 / wsz;
        }
        if (hi >= lo)
            for (int bin = lo; bin <= hi; ++bin)  // x 1/4 (exact): the kernels keep FOUR times the power in LDS
                wr[bin - 4 * pc0] = 0.25 * mel_filters[int64_t(f) * (nf + 2) + (bin - lo)];
    }
    // twiddles, from the same long-double formula as the plan's W_N table
    const long double w = -2.0L * 3.14159265358979323846264338327950288L / (long double)N;
    auto tw = [&](int k, double* out) { out[0] = double(cosl(w * (k % N))); out[1] = double(sinl(w * (k % N))); };
    std::vector<double> twa, tws, gtab;
    std::vector<uint16_t> pairs;
    if (kind != 4) {
        twa.resize(size_t(g.k1_rows - 1) * g.lanes_per_frame * 2);
        tws.resize(size_t(g.split_count) * 2);
        for (int k1 = 1; k1 < g.k1_rows; ++k1)
            for (int j = 0; j < g.lanes_per_frame; ++j) tw(2 * j * k1, &twa[(size_t(k1 - 1) * g.lanes_per_frame + j) * 2]);
        for (int k = 0; k < g.split_count; ++k) tw(k, &tws[size_t(k) * 2]);
    } else {
        // w64x16 (melspec_wave.hip): pass-2 twiddles W_64^(n3 k2) = W_2048^(32 n3 k2) in the blob; the column pairs of
        // every lane; and in GLOBAL memory, lane-ordered: pass-1 twiddles W_1024^(l k1) [15][64] and the split twiddles
        // W_2048^k of the lane's 2 x 5 pairs [2][5][64]
        twa.resize(4 * 16 * 2);
        for (int n3 = 0; n3 < 4; ++n3)
            for (int k2 = 0; k2 < 16; ++k2) tw(32 * n3 * k2, &twa[(size_t(n3) * 16 + k2) * 2]);
        pairs.resize(64 * 4);
        // lane-ordered base twiddles: pass 1 needs W_1024^(l k1), k1 = 1..15 -- the kernel multiplies them together from
        // the four powers k1 = 1, 2, 4, 8 (at most three factors) -- and the split W_2048^ka of the lane's two column
        // pairs (the other pairs' twiddles are that value times an eighth root of unity)
        gtab.resize((4 * 64 + 2 * 64) * 2);
        for (int b = 0; b < 4; ++b)
            for (int l = 0; l < 64; ++l) tw(2 * l * (1 << b), &gtab[(size_t(b) * 64 + l) * 2]);
        for (int l = 0; l < 64; ++l)
            for (int sl = 0; sl < 2; ++sl) {
                // slot q < 127: columns with base bins ka = q + 1 and kb = 256 - ka (its partner column); q = 127: the two
                // self-paired columns 128 and 0
                const int q = l + 64 * sl;
                const bool sp = q == 127;
                const int ka = sp ? 128 : q + 1, kb = sp ? 0 : 255 - q;
                pairs[4 * l + 2 * sl] = uint16_t(ka);
                pairs[4 * l + 2 * sl + 1] = uint16_t(kb);
                tw(ka, &gtab[(size_t(4 * 64) + size_t(sl) * 64 + l) * 2]);
            }
    }
    // the blob
    auto align32 = [](size_t v) { return (v + 31) & ~size_t(31); };
    const size_t w4_bytes = align32(wrows.size() * wsz);
    e.w4_off = 0;
    e.w_stride = int(w_stride);
    e.slots_off = int(w4_bytes);
    e.n_slots = n_slots;
    e.twa_off = int(e.slots_off + align32(slots.size() * 4));
    e.tws_off = int(e.twa_off + align32(twa.size() * tsz));
    e.pairs_off = int(e.tws_off + align32(tws.size() * tsz));
    e.blob_bytes = int(e.pairs_off + align32(pairs.size() * 2));
    e.n_groups = G;
    std::vector<unsigned char> blob(size_t(e.blob_bytes), 0);
    auto put_real = [&](size_t off, const std::vector<double>& v) {
        if (v.empty()) return;
        if (dt == AUD_F64) std::memcpy(&blob[off], v.data(), v.size() * 8);
        else {
            std::vector<float> fv = convert<float>(v.data(), v.size());
            std::memcpy(&blob[off], fv.data(), fv.size() * 4);
        }
    };
    {
        std::vector<float> fw = convert<float>(wrows.data(), wrows.size());
        std::memcpy(&blob[size_t(e.w4_off)], fw.data(), fw.size() * 4);
    }
    std::memcpy(&blob[size_t(e.slots_off)], slots.data(), slots.size() * 4);
    put_real(size_t(e.twa_off), twa);
    put_real(size_t(e.tws_off), tws);
    if (!pairs.empty()) std::memcpy(&blob[size_t(e.pairs_off)], pairs.data(), pairs.size() * 2);
    if (!aud::melspec_wave_finish(kind, dt, &e)) return AUD_OK;  // does not fit LDS
    if (aud::melspec_wave_prepare(kind, dt, &e) != hipSuccess) {
        (void)hipGetLastError();
        return AUD_OK;
    }
    int rc = upload(c, &p->d_blob, blob.data(), blob.size());
    if (rc != AUD_OK) return rc;
    if (!gtab.empty()) {
        if (dt == AUD_F64) rc = upload(c, &p->d_gtab, gtab.data(), gtab.size() * 8);
        else {
            std::vector<float> fv = convert<float>(gtab.data(), gtab.size());
            rc = upload(c, &p->d_gtab, fv.data(), fv.size() * 4);
        }
        if (rc != AUD_OK) return rc;
    }
    e.gtab = p->d_gtab;
    e.blob = p->d_blob;
    p->wv = e;
    p->wave_kind = kind;
    p->use_wave = true;
    p->family = plan_family(p);
    return AUD_OK;
}

const char* plan_family(const aud_plan* p) {
    if (!p->use_wave || !p->wave_kind) return "generic";
    return p->wave_kind == 1 ? "w16x16" : p->wave_kind == 3 ? "w20x10" : "w64x16";
}

}  // namespace

extern "C" {

int aud_init(int device_id, aud_ctx** out) {
    if (!out) return AUD_EINVAL;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return AUD_EHIP;
    aud_ctx* c = new (std::nothrow) aud_ctx();
    if (!c) return AUD_ENOMEM;
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return AUD_EHIP;
    }
    *out = c;
    return AUD_OK;
}

int aud_shutdown(aud_ctx* c) {
    if (!c) return AUD_EINVAL;
    (void)hipSetDevice(c->device);
    aud_comm_destroy(c);
    aud_gather_destroy(c);
    for (int i = 0; i < 4; ++i)
        if (c->ws[i]) (void)hipFree(c->ws[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->rccl_lib) dlclose(c->rccl_lib);
    delete c;
    return AUD_OK;
}

const char* aud_last_error(const aud_ctx* c) {
    if (!c) return "null context";
    static thread_local std::string copy;  // valid until this thread's next call
    std::lock_guard<std::mutex> lk(const_cast<aud_ctx*>(c)->err_mutex);
    copy = c->err;
    return copy.c_str();
}
int aud_device_id(const aud_ctx* c) { return c ? c->device : -1; }

int aud_plan_create(aud_ctx* c, const aud_plan_desc* d, const int32_t* bin_pts, const double* mel_filters,
                    const double* gabor_filters, aud_plan** out) {
    if (!c || !d || !out) return AUD_EINVAL;
    *out = nullptr;
    const int N = d->win_samples, nf = d->mel.n_filters;
    if (N < 4 || d->step_samples < 1 || d->segment_steps < 1 || d->border_steps < 0)
        return fail(c, AUD_EINVAL, "win_samples >= 4, step_samples >= 1, segment_steps >= 1 required");
    if (nf < 1 || !bin_pts || !mel_filters) return fail(c, AUD_EINVAL, "mel table missing");
    if (d->compute_dtype != AUD_F32 && d->compute_dtype != AUD_F64)
        return fail(c, AUD_EINVAL, "compute_dtype must be AUD_F32 or AUD_F64");
    const int H = N / 2 + 1;
    // Envelope of mel.FilterDft (mel.go:128-131): every tap must stay inside Power [H] and
    // inside the [nf, nf+2] table (flat offset); outside it the Go code panics.
    const int64_t cells = int64_t(nf) * (nf + 2);
    for (int f = 0; f < nf; ++f) {
        const int lo = bin_pts[f], hi = bin_pts[f + 2];
        if (lo < 0 || hi >= H) return fail(c, AUD_EINVAL, "mel BinPts outside the power spectrum (HiHz > Nyquist?)");
        if (hi >= lo && int64_t(f) * (nf + 2) + (hi - lo) >= cells)
            return fail(c, AUD_EINVAL, "mel triangle wider than the filter table (SURVEY Q4)");
    }
    if (d->mfcc_coefs < 0 || d->mfcc_coefs > nf || (d->mfcc_coefs > 0 && nf < 2))
        return fail(c, AUD_EINVAL, "mfcc_coefs must be 0..n_filters (and n_filters >= 2: fourier.NewDCT panics)");
    if (d->n_gabor > 0) {
        if (!gabor_filters || d->gabor.size_x < 1 || d->gabor.size_y < 1 || d->gabor.stride_x < 1 ||
            d->gabor.stride_y < 1)
            return fail(c, AUD_EINVAL, "gabor filter set incomplete");
    }
    AUD_HIP(c, make_current(c));

    aud_plan* p = new (std::nothrow) aud_plan();
    if (!p) return AUD_ENOMEM;
    p->ctx = c;
    p->d = *d;
    p->H = H;
    p->ratio = (N % 2 == 0) ? 2 : 1;
    p->M = N / p->ratio;
    factorize(p->M, p->fac, &p->nfac);
    p->F_generic = aud::melspec_generic_pick_F(p->M, d->compute_dtype);
    if (p->F_generic < 1) {
        delete p;
        return fail(c, AUD_EINVAL, "win_samples too large for the LDS-resident FFT");
    }

    int rc = AUD_OK;
    {  // twiddles exp(-2 pi i k / N), computed in long double
        std::vector<double> tw(size_t(N) * 2);
        const long double w = -2.0L * 3.14159265358979323846264338327950288L / (long double)N;
        for (int k = 0; k < N; ++k) {
            tw[2 * size_t(k)] = double(cosl(w * k));
            tw[2 * size_t(k) + 1] = double(sinl(w * k));
        }
        rc = upload_real(c, &p->d_tw, tw.data(), tw.size(), d->compute_dtype);
    }
    // A prime factor the register radices do not cover costs O(p) per output (N = 1103, what 25 ms at 44.1 kHz gives, is
    // prime: 0.6 M complex multiply-adds per frame).  Such lengths go through Bluestein's chirp convolution instead: two
    // power-of-two FFTs of length L >= 2 M - 1 (melspec_generic.hip).  Tables in long double.
    {
        bool awkward = false;
        for (int i = 0; i < p->nfac; ++i) awkward = awkward || p->fac[i] > 25;
        const int L = awkward ? aud::melspec_generic_bluestein_L(p->M, d->compute_dtype) : 0;
        if (rc == AUD_OK && L > 0) {
            const int M = p->M;
            const long double pi = 3.14159265358979323846264338327950288L;
            const size_t Mz = size_t(M), Lz = size_t(L);
            std::vector<long double> wr(Mz), wi(Mz);
            for (int n = 0; n < M; ++n) {  // exp(-i pi n^2 / M) with n^2 reduced mod 2 M
                const long double ang = -pi * (long double)((int64_t(n) * n) % (2 * int64_t(M))) / (long double)M;
                wr[size_t(n)] = cosl(ang);
                wi[size_t(n)] = sinl(ang);
            }
            // b[m] = conj(w[|m|]) wrapped to length L, bhat = FFT_L(b) / L by an iterative radix-2 FFT in long double
            std::vector<long double> br(Lz, 0.0L), bi(Lz, 0.0L);
            for (int m = 0; m < M; ++m) {
                br[size_t(m)] = wr[size_t(m)];
                bi[size_t(m)] = -wi[size_t(m)];
                if (m > 0) {
                    br[size_t(L - m)] = wr[size_t(m)];
                    bi[size_t(L - m)] = -wi[size_t(m)];
                }
            }
            for (int i = 1, j = 0; i < L; ++i) {  // bit reversal
                int bit = L >> 1;
                for (; j & bit; bit >>= 1) j ^= bit;
                j ^= bit;
                if (i < j) {
                    std::swap(br[size_t(i)], br[size_t(j)]);
                    std::swap(bi[size_t(i)], bi[size_t(j)]);
                }
            }
            for (int len = 2; len <= L; len <<= 1) {
                const long double ang = -2.0L * pi / (long double)len;
                for (int i0 = 0; i0 < L; i0 += len)
                    for (int k = 0; k < len / 2; ++k) {
                        const long double cr = cosl(ang * k), ci = sinl(ang * k);
                        const size_t u = size_t(i0 + k), v = size_t(i0 + k + len / 2);
                        const long double tr = br[v] * cr - bi[v] * ci, ti = br[v] * ci + bi[v] * cr;
                        br[v] = br[u] - tr;
                        bi[v] = bi[u] - ti;
                        br[u] += tr;
                        bi[u] += ti;
                    }
            }
            std::vector<double> chirp(Mz * 2), bhat(Lz * 2), twl(Lz * 2);
            for (int n = 0; n < M; ++n) {
                chirp[2 * size_t(n)] = double(wr[size_t(n)]);
                chirp[2 * size_t(n) + 1] = double(wi[size_t(n)]);
            }
            for (int k = 0; k < L; ++k) {
                bhat[2 * size_t(k)] = double(br[size_t(k)] / (long double)L);
                bhat[2 * size_t(k) + 1] = double(bi[size_t(k)] / (long double)L);
                const long double ang = -2.0L * pi * k / (long double)L;
                twl[2 * size_t(k)] = double(cosl(ang));
                twl[2 * size_t(k) + 1] = double(sinl(ang));
            }
            rc = upload_real(c, &p->d_bl_chirp, chirp.data(), chirp.size(), d->compute_dtype);
            if (rc == AUD_OK) rc = upload_real(c, &p->d_bl_bhat, bhat.data(), bhat.size(), d->compute_dtype);
            if (rc == AUD_OK) rc = upload_real(c, &p->d_bl_tw, twl.data(), twl.size(), d->compute_dtype);
            const size_t lds = aud::melspec_generic_lds_bytes(L, 1, d->compute_dtype);
            if (rc == AUD_OK && lds > 64u * 1024u && aud::melspec_generic_prepare(lds) != hipSuccess) {
                (void)hipGetLastError();
                rc = fail(c, AUD_EHIP, "the runtime refused the LDS size of the Bluestein route");
            }
            if (rc == AUD_OK) {
                p->bl_L = L;
                factorize(L, p->bl_fac, &p->bl_nfac);
                p->F_generic = 1;
            }
        }
    }
    if (rc == AUD_OK) rc = upload_real(c, &p->d_filt, mel_filters, size_t(cells), d->compute_dtype);
    if (rc == AUD_OK)
        rc = upload(c, reinterpret_cast<void**>(&p->d_bin_pts), bin_pts, sizeof(int32_t) * (nf + 2));
    if (rc == AUD_OK && d->n_gabor > 0)
        rc = upload_real(c, &p->d_gabor, gabor_filters,
                         size_t(d->n_gabor) * d->gabor.size_x * d->gabor.size_y, d->compute_dtype);
    if (rc == AUD_OK && d->mfcc_coefs > 0) {
        // rows of the unnormalised DCT-I (FFTPACK cost / gonum fourier.DCT):
        // y[k] = x[0] + (-1)^k x[n-1] + 2 sum_{j=1}^{n-2} x[j] cos(pi j k / (n-1))
        std::vector<double> dct(size_t(d->mfcc_coefs) * nf);
        const long double pi = 3.14159265358979323846264338327950288L;
        for (int k = 0; k < d->mfcc_coefs; ++k)
            for (int j = 0; j < nf; ++j) {
                double v;
                if (j == 0) v = 1.0;
                else if (j == nf - 1) v = (k & 1) ? -1.0 : 1.0;
                else v = double(2.0L * cosl(pi * (long double)j * (long double)k / (long double)(nf - 1)));
                dct[size_t(k) * nf + j] = v;
            }
        rc = upload_real(c, &p->d_dct, dct.data(), dct.size(), d->compute_dtype);
    }
    // kernel family: the wave-autonomous kernel of this window length where one exists and its tables fit, else the
    // generic any-N kernel
    if (rc == AUD_OK) rc = build_wave_tables(p, bin_pts, mel_filters);
    if (rc != AUD_OK) {
        aud_plan_destroy(p);
        return rc;
    }
    *out = p;
    return AUD_OK;
}

int aud_plan_destroy(aud_plan* p) {
    if (!p) return AUD_EINVAL;
    (void)hipSetDevice(p->ctx->device);
    if (p->d_tw) (void)hipFree(p->d_tw);
    if (p->d_bl_chirp) (void)hipFree(p->d_bl_chirp);
    if (p->d_bl_bhat) (void)hipFree(p->d_bl_bhat);
    if (p->d_bl_tw) (void)hipFree(p->d_bl_tw);
    if (p->d_filt) (void)hipFree(p->d_filt);
    if (p->d_bin_pts) (void)hipFree(p->d_bin_pts);
    if (p->d_gabor) (void)hipFree(p->d_gabor);
    if (p->d_dct) (void)hipFree(p->d_dct);
    if (p->d_blob) (void)hipFree(p->d_blob);
    if (p->d_gtab) (void)hipFree(p->d_gtab);
    delete p;
    return AUD_OK;
}

const char* aud_plan_kernel_name(const aud_plan* p) { return p ? p->family : ""; }

int aud_plan_get_info(const aud_plan* p, const char* name, int64_t* value) {
    if (!p || !name || !value) return AUD_EINVAL;
    const std::string key(name);
    const bool wave = p->use_wave && p->wave_kind;
    if (key == "lds_bytes") *value = wave ? int64_t(p->wv.lds_bytes) : 0;
    else if (key == "waves_per_wg") *value = wave ? p->wv.waves : 4;
    else if (key == "wgs_per_cu") *value = wave ? p->wv.wgs_per_cu : 0;
    else if (key == "bluestein_L") *value = wave ? 0 : p->bl_L;
    else if (key == "frames_per_wave") *value = wave ? aud::melspec_wave_frames_per_wave(p->wave_kind) : 0;
    else if (key == "epilogue_steps") {
        int n = 0;
        for (int k = 0; k < p->wv.n_slots && k < 8; ++k) n += p->wv.slot_steps[k];
        *value = wave ? n : 0;
    } else return fail(p->ctx, AUD_EINVAL, "aud_plan_get_info: unknown name");
    return AUD_OK;
}

int aud_plan_set_option(aud_plan* p, const char* name, int value) {
    if (!p || !name) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    const std::string key(name);
    if (key == "kernel") {  // 0 = automatic choice, 1 = the generic any-N kernel
        if (value < 0 || value > 1) return fail(c, AUD_EINVAL, "kernel: 0 (auto) or 1 (generic)");
        p->use_wave = value == 0 && p->wave_kind != 0;
        p->family = plan_family(p);
        return AUD_OK;
    }
    if (key == "xcd_remap") {  // 1 (default): every XCD walks a contiguous run of tiles; 0: tiles in workgroup-id order
        if (value != 0 && value != 1) return fail(c, AUD_EINVAL, "xcd_remap: 0 or 1");
        p->xcd_remap = value;
        return AUD_OK;
    }
#ifdef AUD_STAMPS
    if (key == "stamps_lo") { p->stamps = (p->stamps & 0xFFFFFFFF00000000ull) | uint32_t(value); return AUD_OK; }
    if (key == "stamps_hi") { p->stamps = (p->stamps & 0xFFFFFFFFull) | (uint64_t(uint32_t(value)) << 32); return AUD_OK; }
#endif
    return fail(c, AUD_EINVAL, "unknown option");
}

int aud_melspec_batch_dev(aud_plan* p, const void* sig, int sig_dtype, const aud_item* items,
                          int n_items, float* mel, float* power, float* log_power, void* stream) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (n_items < 0 || (n_items > 0 && (!sig || !items || !mel)))
        return fail(c, AUD_EINVAL, "null buffer");
    if (sig_dtype != AUD_F32 && sig_dtype != AUD_F64 && sig_dtype != AUD_I16)
        return fail(c, AUD_EINVAL, "bad sig_dtype");
    if (log_power && !p->d.dft.comp_log_pow) return fail(c, AUD_EINVAL, "log_power needs CompLogPow");
    const bool smooth = p->d.dft.prev_smooth != 0.0;
    if (smooth && !power)
        return fail(c, AUD_EINVAL, "dft.PrevSmooth != 0 needs the power buffer (the scan runs on it)");
    if (n_items == 0) return AUD_OK;
    // one workgroup per (item, frame tile): keep the 1-D grid inside what a launch accepts
    if (int64_t(n_items) * int64_t(p->d.segment_steps) > (int64_t(1) << 30))
        return fail(c, AUD_EINVAL, "n_items x segment_steps too large for one launch; split the batch");
    AUD_HIP(c, make_current(c));
    aud::MelspecArgs a;
    fill_melspec_args(p, &a);
    a.sig = sig;
    a.sig_dtype = sig_dtype;
    a.items = items;
    a.n_items = n_items;
    a.mel = mel;
    a.power = power;
    a.log_power = log_power;
    AUD_HIP(c, launch_frames(p, a, static_cast<hipStream_t>(stream)));
    if (smooth) {
        // dft.go:67-69: p_s = Prev*p_{s-1} + Cur*raw_s along the steps, then log-power and mel from it
        aud::SmoothArgs sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.items = items;
        sa.n_items = n_items;
        sa.H = p->H;
        sa.T = p->d.segment_steps;
        sa.N = p->d.win_samples;
        sa.S = p->d.step_samples;
        sa.border = p->d.border_steps;
        sa.power = power;
        sa.log_power = log_power;
        sa.prev_smooth = p->d.dft.prev_smooth;
        sa.cur_smooth = p->d.dft.cur_smooth;
        sa.log_off = p->d.dft.log_offset;
        sa.log_min = p->d.dft.log_min;
        sa.comp_log_pow = p->d.dft.comp_log_pow;
        AUD_HIP(c, aud::launch_power_smooth(sa, p->d.compute_dtype, static_cast<hipStream_t>(stream)));
        AUD_HIP(c, aud::launch_mel_from_power(a, p->d.compute_dtype, static_cast<hipStream_t>(stream)));
    }
    return AUD_OK;
}

int aud_mfcc_batch_dev(aud_plan* p, const aud_item* items, int n_items, const float* mel, const float* log_power,
                       float* mfcc, float* deltas, float* delta_deltas, float* energy, void* stream) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (p->d.mfcc_coefs <= 0 || !p->d_dct) return fail(c, AUD_EINVAL, "plan was created without mfcc_coefs");
    if (n_items < 0) return fail(c, AUD_EINVAL, "bad n_items");
    if (p->d.segment_steps > p->H)
        return fail(c, AUD_EINVAL, "Energy reads LogPowerSegment row s < SegmentSteps: needs SegmentSteps <= bins (Go panics)");
    if (delta_deltas && !deltas) return fail(c, AUD_EINVAL, "delta_deltas needs deltas");
    if (n_items == 0) return AUD_OK;
    if (!items || !mel || !log_power || !mfcc) return fail(c, AUD_EINVAL, "null buffer");
    AUD_HIP(c, make_current(c));
    aud::MfccArgs a;
    std::memset(&a, 0, sizeof(a));
    a.items = items;
    a.n_items = n_items;
    a.N = p->d.win_samples;
    a.S = p->d.step_samples;
    a.T = p->d.segment_steps;
    a.border = p->d.border_steps;
    a.H = p->H;
    a.nf = p->d.mel.n_filters;
    a.n_coefs = p->d.mfcc_coefs;
    a.dct = p->d_dct;
    a.mel = mel;
    a.log_power = log_power;
    a.mfcc = mfcc;
    a.deltas = deltas;
    a.delta_deltas = delta_deltas;
    a.energy = energy;
    AUD_HIP(c, aud::launch_mfcc(a, p->d.compute_dtype, static_cast<hipStream_t>(stream)));
    return AUD_OK;
}

int aud_gabor_batch_dev(aud_plan* p, const float* mel, int n_items, int rows, int cols, int out_rank,
                        const int32_t* out_shape, int by_time, float* out, void* stream) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (p->d.n_gabor <= 0 || !p->d_gabor) return fail(c, AUD_EINVAL, "plan has no gabor filters");
    if (n_items < 0 || rows < 1 || cols < 1 || !out_shape) return fail(c, AUD_EINVAL, "bad shape");
    const aud_gabor_set& g = p->d.gabor;
    int32_t nT = 0, nF = 0, strides = 1;
    if (aud_gabor_iter_space(&g, rows, cols, out_rank, out_shape, &nT, &nF, &strides) != AUD_OK)
        return fail(c, AUD_EINVAL, "Convolve rejects this shape (gabor.go:226-229, :259-262)");
    if (n_items == 0 || nT == 0 || nF == 0) return AUD_OK;
    if (!mel || !out) return fail(c, AUD_EINVAL, "null buffer");
    const int nG = p->d.n_gabor;
    // reads: the reference indexes melData by flat offset; past the end it panics
    const int64_t last_read = int64_t((nF - 1) * g.stride_y + g.size_y - 1) * cols +
                              int64_t(nT - 1) * g.stride_x + g.size_x - 1;
    if (last_read >= int64_t(rows) * cols)
        return fail(c, AUD_EINVAL, "gabor pools reach past the mel matrix (SURVEY Q10)");
    aud::GaborArgs a;
    std::memset(&a, 0, sizeof(a));
    if (out_rank == 2) {
        const int x_max = by_time ? (nT - 1) + strides * (nG - 1) : (nG - 1) + (nT - 1) * nG;
        if (2 * nF > out_shape[0] || x_max >= out_shape[1])
            return fail(c, AUD_EINVAL, "2-D gabor output too small");
        a.d0 = out_shape[0];
        a.d1 = out_shape[1];
    } else {
        if (nF > out_shape[0] || nT > out_shape[1] || out_shape[2] < 2 || out_shape[3] < nG)
            return fail(c, AUD_EINVAL, "4-D gabor output too small");
        a.d0 = out_shape[0];
        a.d1 = out_shape[1];
        a.d2 = out_shape[2];
        a.d3 = out_shape[3];
    }
    AUD_HIP(c, make_current(c));
    a.mel = mel;
    a.n_items = n_items;
    a.rows = rows;
    a.cols = cols;
    a.k = p->d_gabor;
    a.nG = nG;
    a.SX = g.size_x;
    a.SY = g.size_y;
    a.stx = g.stride_x;
    a.sty = g.stride_y;
    a.gain = g.gain;
    a.rank = out_rank;
    a.by_time = by_time;
    a.nT = nT;
    a.nF = nF;
    a.t_max_strides = strides;
    a.out = out;
    AUD_HIP(c, aud::launch_gabor(a, p->d.compute_dtype, static_cast<hipStream_t>(stream)));
    return AUD_OK;
}

int aud_process_batch_dev(aud_plan* p, const void* sig, int sig_dtype, const aud_item* items,
                          int n_items, float* mel, int pools_y, int pools_x, float* gabor,
                          void* stream) {
    if (!p) return AUD_EINVAL;
    int rc = aud_melspec_batch_dev(p, sig, sig_dtype, items, n_items, mel, nullptr, nullptr, stream);
    if (rc != AUD_OK) return rc;
    const int32_t shape[4] = {pools_y, pools_x, 2, p->d.n_gabor};
    return aud_gabor_batch_dev(p, mel, n_items, p->d.mel.n_filters, p->d.segment_steps, 4, shape, 0,
                               gabor, stream);
}

int aud_melspec_batch_host(aud_plan* p, const double* sig, int64_t sig_total, const aud_item* items,
                           int n_items, double* mel, double* power, double* log_power) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (n_items < 0 || sig_total < 0 || (n_items > 0 && (!sig || !items || !mel)))
        return fail(c, AUD_EINVAL, "null buffer");
    if (n_items == 0) return AUD_OK;
    for (int i = 0; i < n_items; ++i)
        if (items[i].sig_off < 0 || items[i].sig_len < 0 || items[i].sig_stride < 0 || item_last(items[i]) >= sig_total)
            return fail(c, AUD_EINVAL, "item outside the signal buffer");
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    const int nf = p->d.mel.n_filters, T = p->d.segment_steps, H = p->H;
    const size_t n_mel = size_t(n_items) * nf * T, n_pow = size_t(n_items) * H * T;
    const size_t sig_bytes = size_t(sig_total) * 8, item_bytes = size_t(n_items) * sizeof(aud_item);
    const bool smooth = p->d.dft.prev_smooth != 0.0;  // the scan needs a device power buffer
    const bool want_p = power != nullptr || smooth, want_lp = log_power != nullptr;
    const size_t out_floats = n_mel + (want_p ? n_pow : 0) + (want_lp ? n_pow : 0);
    int rc;
    if ((rc = ensure_ws(c, 0, sig_bytes + 16)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 1, item_bytes)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 2, out_floats * 4)) != AUD_OK) return rc;
    float* d_mel = static_cast<float*>(c->ws[2]);
    float* d_pow = want_p ? d_mel + n_mel : nullptr;
    float* d_lp = want_lp ? d_mel + n_mel + (want_p ? n_pow : 0) : nullptr;
    AUD_HIP(c, hipMemcpyAsync(c->ws[0], sig, sig_bytes, hipMemcpyHostToDevice, c->stream));
    AUD_HIP(c, hipMemcpyAsync(c->ws[1], items, item_bytes, hipMemcpyHostToDevice, c->stream));
    rc = aud_melspec_batch_dev(p, c->ws[0], AUD_F64, static_cast<const aud_item*>(c->ws[1]), n_items,
                               d_mel, d_pow, d_lp, c->stream);
    if (rc != AUD_OK) return rc;
    std::vector<float> h(out_floats);
    AUD_HIP(c, hipMemcpyAsync(h.data(), d_mel, out_floats * 4, hipMemcpyDeviceToHost, c->stream));
    AUD_HIP(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n_mel; ++i) mel[i] = double(h[i]);
    size_t o = n_mel;
    if (want_p) {
        if (power)
            for (size_t i = 0; i < n_pow; ++i) power[i] = double(h[o + i]);
        o += n_pow;
    }
    if (want_lp)
        for (size_t i = 0; i < n_pow; ++i) log_power[i] = double(h[o + i]);
    return AUD_OK;
}

int aud_snd_to_window(const double* signal, int64_t sig_len, int64_t start, int win_samples, double* window) {
    if (!signal || !window || win_samples < 1 || sig_len < 0) return AUD_EINVAL;
    const int64_t end = start + win_samples;
    if (end > sig_len) return AUD_ESHORT;  // "SndToWindow: end beyond signal length!!"
    for (int64_t i = 0; i < win_samples; ++i) {
        const int64_t pos = start + i;
        window[i] = pos < 0 ? 0.0 : signal[pos];
    }
    return AUD_OK;
}

int aud_dft_filter_host(aud_plan* p, int step, const double* window, double* power, double* log_power,
                        double* power_seg, double* log_power_seg) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    const int N = p->d.win_samples, T = p->d.segment_steps, H = p->H, nf = p->d.mel.n_filters;
    if (!window || !power || !power_seg || step < 0 || step >= T) return fail(c, AUD_EINVAL, "bad argument");
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    // the window becomes a one-frame stream: frame 0 of the item covers [0, N), every later frame is dead
    const aud_item it{0, N, p->d.step_samples * p->d.border_steps};
    const size_t n_mel = size_t(nf) * T, n_pow = size_t(H) * T;
    int rc;
    if ((rc = ensure_ws(c, 0, size_t(N) * 8 + 16)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 1, sizeof(aud_item) + size_t(H) * 8 * 3)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 2, (n_mel + n_pow) * 4)) != AUD_OK) return rc;
    unsigned char* w1 = static_cast<unsigned char*>(c->ws[1]);
    double* d_carry = reinterpret_cast<double*>(w1 + sizeof(aud_item));
    double* d_p = d_carry + H;
    double* d_lp = d_p + H;
    float* d_mel = static_cast<float*>(c->ws[2]);
    float* d_pow = d_mel + n_mel;
    AUD_HIP(c, hipMemcpyAsync(c->ws[0], window, size_t(N) * 8, hipMemcpyHostToDevice, c->stream));
    AUD_HIP(c, hipMemcpyAsync(w1, &it, sizeof(it), hipMemcpyHostToDevice, c->stream));
    AUD_HIP(c, hipMemcpyAsync(d_carry, power, size_t(H) * 8, hipMemcpyHostToDevice, c->stream));
    aud::MelspecArgs a;
    fill_melspec_args(p, &a);
    a.sig = c->ws[0];
    a.sig_dtype = AUD_F64;
    a.items = reinterpret_cast<const aud_item*>(w1);
    a.n_items = 1;
    a.mel = d_mel;
    a.power = d_pow;
    AUD_HIP(c, launch_frames(p, a, c->stream));
    AUD_HIP(c, aud::launch_frame_blend(d_pow, T, d_carry, H, step, p->d.dft.prev_smooth, p->d.dft.cur_smooth,
                                       p->d.dft.comp_log_pow, p->d.dft.log_offset, p->d.dft.log_min, d_p, d_lp,
                                       p->d.compute_dtype, c->stream));
    std::vector<double> hp(size_t(H) * 2);
    AUD_HIP(c, hipMemcpyAsync(hp.data(), d_p, size_t(H) * 16, hipMemcpyDeviceToHost, c->stream));
    AUD_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < H; ++k) {  // the tensor stores of dft.go:70-83
        power[k] = hp[k];
        power_seg[size_t(k) * T + step] = hp[k];
        if (p->d.dft.comp_log_pow) {
            if (log_power) log_power[k] = hp[size_t(H) + k];
            if (log_power_seg) log_power_seg[size_t(k) * T + step] = hp[size_t(H) + k];
        }
    }
    return AUD_OK;
}

int aud_dft_power_host(aud_plan* p, int step, const double* fft_coefs, double* power, double* log_power,
                       double* power_seg, double* log_power_seg) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    const int T = p->d.segment_steps, H = p->H;
    if (!fft_coefs || !power || !power_seg || step < 0 || step >= T) return fail(c, AUD_EINVAL, "bad argument");
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    int rc;
    if ((rc = ensure_ws(c, 0, size_t(H) * 16 + 16)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 1, size_t(H) * 8 * 3 + size_t(H) * 4 + 16)) != AUD_OK) return rc;
    double* d_carry = static_cast<double*>(c->ws[1]);
    double* d_p = d_carry + H;
    double* d_lp = d_p + H;
    float* d_raw = reinterpret_cast<float*>(d_lp + H);
    AUD_HIP(c, hipMemcpyAsync(c->ws[0], fft_coefs, size_t(H) * 16, hipMemcpyHostToDevice, c->stream));
    AUD_HIP(c, hipMemcpyAsync(d_carry, power, size_t(H) * 8, hipMemcpyHostToDevice, c->stream));
    AUD_HIP(c, aud::launch_power_from_coefs(static_cast<const double*>(c->ws[0]), H, d_raw, p->d.compute_dtype,
                                            c->stream));
    AUD_HIP(c, aud::launch_frame_blend(d_raw, 1, d_carry, H, step, p->d.dft.prev_smooth, p->d.dft.cur_smooth,
                                       p->d.dft.comp_log_pow, p->d.dft.log_offset, p->d.dft.log_min, d_p, d_lp,
                                       p->d.compute_dtype, c->stream));
    std::vector<double> hp(size_t(H) * 2);
    AUD_HIP(c, hipMemcpyAsync(hp.data(), d_p, size_t(H) * 16, hipMemcpyDeviceToHost, c->stream));
    AUD_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < H; ++k) {  // the tensor stores of dft.go:70-83
        power[k] = hp[k];
        power_seg[size_t(k) * T + step] = hp[k];
        if (p->d.dft.comp_log_pow) {
            if (log_power) log_power[k] = hp[size_t(H) + k];
            if (log_power_seg) log_power_seg[size_t(k) * T + step] = hp[size_t(H) + k];
        }
    }
    return AUD_OK;
}

int aud_cepstrum_dct_host(aud_plan* p, int step, const double* fbank, double* mfcc_seg, double* mfcc_dct) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    const int N = p->d.win_samples, T = p->d.segment_steps, nf = p->d.mel.n_filters, nc = p->d.mfcc_coefs;
    if (nc < 1 || !p->d_dct) return fail(c, AUD_EINVAL, "plan was created without mfcc_coefs");
    if (!fbank || !mfcc_seg || step < 0 || step >= T) return fail(c, AUD_EINVAL, "bad argument");
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    int rc;
    if ((rc = ensure_ws(c, 1, sizeof(aud_item))) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 2, size_t(nf + nc) * 4 + 16)) != AUD_OK) return rc;
    const aud_item it{0, N, p->d.step_samples * p->d.border_steps};  // a one-step segment whose only step is live
    float* d_mel = static_cast<float*>(c->ws[2]);
    float* d_mfcc = d_mel + nf;
    std::vector<float> hm(static_cast<size_t>(nf));
    for (int j = 0; j < nf; ++j) hm[size_t(j)] = float(fbank[j]);
    AUD_HIP(c, hipMemcpyAsync(c->ws[1], &it, sizeof(it), hipMemcpyHostToDevice, c->stream));
    AUD_HIP(c, hipMemcpyAsync(d_mel, hm.data(), size_t(nf) * 4, hipMemcpyHostToDevice, c->stream));
    aud::MfccArgs a;
    std::memset(&a, 0, sizeof(a));
    a.items = static_cast<const aud_item*>(c->ws[1]);
    a.n_items = 1;
    a.N = N;
    a.S = p->d.step_samples;
    a.T = 1;
    a.border = p->d.border_steps;
    a.H = p->H;
    a.nf = nf;
    a.n_coefs = nc;
    a.dct = p->d_dct;
    a.mel = d_mel;
    a.mfcc = d_mfcc;
    AUD_HIP(c, aud::launch_mfcc_dct(a, p->d.compute_dtype, c->stream));
    std::vector<float> out(static_cast<size_t>(nc));
    AUD_HIP(c, hipMemcpyAsync(out.data(), d_mfcc, size_t(nc) * 4, hipMemcpyDeviceToHost, c->stream));
    AUD_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < nc; ++i) mfcc_seg[size_t(i) * T + step] = double(out[size_t(i)]);  // mel.go:207-209
    if (mfcc_dct)
        for (int j = 0; j < nf; ++j) mfcc_dct[j] = fbank[j];  // mel.go:193: the work tensor ends up a copy of the input
    return AUD_OK;
}

int aud_mel_filter_dft_host(aud_plan* p, int step, const double* power, double* segment, double* fbank) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    const int N = p->d.win_samples, T = p->d.segment_steps, H = p->H, nf = p->d.mel.n_filters;
    if (!power || !segment || step < 0 || step >= T) return fail(c, AUD_EINVAL, "bad argument");
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    const aud_item it{0, N, p->d.step_samples * p->d.border_steps};  // only column 0 is live
    const size_t n_mel = size_t(nf) * T, n_pow = size_t(H) * T;
    int rc;
    if ((rc = ensure_ws(c, 1, sizeof(aud_item))) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 2, (n_mel + n_pow) * 4)) != AUD_OK) return rc;
    float* d_mel = static_cast<float*>(c->ws[2]);
    float* d_pow = d_mel + n_mel;
    std::vector<float> hpow(n_pow, 0.f);
    for (int k = 0; k < H; ++k) hpow[size_t(k) * T] = float(power[k]);
    AUD_HIP(c, hipMemcpyAsync(c->ws[1], &it, sizeof(it), hipMemcpyHostToDevice, c->stream));
    AUD_HIP(c, hipMemcpyAsync(d_pow, hpow.data(), n_pow * 4, hipMemcpyHostToDevice, c->stream));
    aud::MelspecArgs a;
    fill_melspec_args(p, &a);
    a.items = static_cast<const aud_item*>(c->ws[1]);
    a.n_items = 1;
    a.mel = d_mel;
    a.power = d_pow;
    AUD_HIP(c, aud::launch_mel_from_power(a, p->d.compute_dtype, c->stream));
    std::vector<float> hm(n_mel);
    AUD_HIP(c, hipMemcpyAsync(hm.data(), d_mel, n_mel * 4, hipMemcpyDeviceToHost, c->stream));
    AUD_HIP(c, hipStreamSynchronize(c->stream));
    for (int f = 0; f < nf; ++f) {  // mel.go:150-151
        const double v = double(hm[size_t(f) * T]);
        if (fbank) fbank[f] = v;
        segment[size_t(f) * T + step] = v;
    }
    return AUD_OK;
}

int aud_melspec_mfcc_batch_host(aud_plan* p, const double* sig, int64_t sig_total, const aud_item* items,
                                int n_items, double* mel, double* power, double* log_power, double* mfcc,
                                double* deltas, double* delta_deltas, double* energy) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (p->d.mfcc_coefs <= 0) return fail(c, AUD_EINVAL, "plan was created without mfcc_coefs");
    if (!p->d.dft.comp_log_pow) return fail(c, AUD_EINVAL, "the MFCC tail reads LogPowerSegment: needs CompLogPow");
    if (n_items < 0 || sig_total < 0 || (n_items > 0 && (!sig || !items || !mel || !mfcc)))
        return fail(c, AUD_EINVAL, "null buffer");
    if (delta_deltas && !deltas) return fail(c, AUD_EINVAL, "delta_deltas needs deltas");
    if (n_items == 0) return AUD_OK;
    for (int i = 0; i < n_items; ++i)
        if (items[i].sig_off < 0 || items[i].sig_len < 0 || items[i].sig_stride < 0 || item_last(items[i]) >= sig_total)
            return fail(c, AUD_EINVAL, "item outside the signal buffer");
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    const int nf = p->d.mel.n_filters, T = p->d.segment_steps, H = p->H, nc = p->d.mfcc_coefs;
    const size_t n_mel = size_t(n_items) * nf * T, n_pow = size_t(n_items) * H * T;
    const size_t n_cc = size_t(n_items) * nc * T, n_en = size_t(n_items) * T;
    // device layout: mel | power | log_power | mfcc | deltas | delta_deltas | energy
    const size_t total = n_mel + 2 * n_pow + 3 * n_cc + n_en;
    const size_t sig_bytes = size_t(sig_total) * 8, item_bytes = size_t(n_items) * sizeof(aud_item);
    int rc;
    if ((rc = ensure_ws(c, 0, sig_bytes + 16)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 1, item_bytes)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 2, total * 4)) != AUD_OK) return rc;
    float* d_mel = static_cast<float*>(c->ws[2]);
    float* d_pow = d_mel + n_mel;
    float* d_lp = d_pow + n_pow;
    float* d_cc = d_lp + n_pow;
    float* d_dl = d_cc + n_cc;
    float* d_ddl = d_dl + n_cc;
    float* d_en = d_ddl + n_cc;
    AUD_HIP(c, hipMemcpyAsync(c->ws[0], sig, sig_bytes, hipMemcpyHostToDevice, c->stream));
    AUD_HIP(c, hipMemcpyAsync(c->ws[1], items, item_bytes, hipMemcpyHostToDevice, c->stream));
    const aud_item* d_items = static_cast<const aud_item*>(c->ws[1]);
    rc = aud_melspec_batch_dev(p, c->ws[0], AUD_F64, d_items, n_items, d_mel, d_pow, d_lp, c->stream);
    if (rc == AUD_OK)
        rc = aud_mfcc_batch_dev(p, d_items, n_items, d_mel, d_lp, d_cc, deltas ? d_dl : nullptr,
                                delta_deltas ? d_ddl : nullptr, d_en, c->stream);
    if (rc != AUD_OK) {
        (void)hipStreamSynchronize(c->stream);
        return rc;
    }
    std::vector<float> h(total);
    AUD_HIP(c, hipMemcpyAsync(h.data(), d_mel, total * 4, hipMemcpyDeviceToHost, c->stream));
    AUD_HIP(c, hipStreamSynchronize(c->stream));
    auto widen = [&](double* dst, const float* src, size_t n) {
        if (dst)
            for (size_t i = 0; i < n; ++i) dst[i] = double(src[i]);
    };
    const float* hp = h.data();
    widen(mel, hp, n_mel);
    widen(power, hp + n_mel, n_pow);
    widen(log_power, hp + n_mel + n_pow, n_pow);
    widen(mfcc, hp + n_mel + 2 * n_pow, n_cc);
    widen(deltas, hp + n_mel + 2 * n_pow + n_cc, n_cc);
    widen(delta_deltas, hp + n_mel + 2 * n_pow + 2 * n_cc, n_cc);
    widen(energy, hp + n_mel + 2 * n_pow + 3 * n_cc, n_en);
    return AUD_OK;
}

int aud_gabor_batch_host(aud_plan* p, const double* mel, int n_items, int rows, int cols, int out_rank,
                         const int32_t* out_shape, int by_time, float* out) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (n_items < 0 || rows < 1 || cols < 1 || !out_shape || (out_rank != 2 && out_rank != 4))
        return fail(c, AUD_EINVAL, "bad shape");
    if (n_items == 0) return AUD_OK;
    if (!mel || !out) return fail(c, AUD_EINVAL, "null buffer");
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    size_t out_cells = 1;
    for (int i = 0; i < out_rank; ++i) out_cells *= size_t(out_shape[i] > 0 ? out_shape[i] : 0);
    const size_t n_mel = size_t(n_items) * rows * cols, n_out = size_t(n_items) * out_cells;
    int rc;
    if ((rc = ensure_ws(c, 2, n_mel * 4)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 3, n_out * 4 + 16)) != AUD_OK) return rc;
    std::vector<float> hm(n_mel);
    for (size_t i = 0; i < n_mel; ++i) hm[i] = float(mel[i]);
    AUD_HIP(c, hipMemcpyAsync(c->ws[2], hm.data(), n_mel * 4, hipMemcpyHostToDevice, c->stream));
    // in/out semantics: cells the reference leaves alone keep the caller's values
    AUD_HIP(c, hipMemcpyAsync(c->ws[3], out, n_out * 4, hipMemcpyHostToDevice, c->stream));
    rc = aud_gabor_batch_dev(p, static_cast<const float*>(c->ws[2]), n_items, rows, cols, out_rank,
                             out_shape, by_time, static_cast<float*>(c->ws[3]), c->stream);
    if (rc != AUD_OK) {
        (void)hipStreamSynchronize(c->stream);
        return rc;
    }
    AUD_HIP(c, hipMemcpyAsync(out, c->ws[3], n_out * 4, hipMemcpyDeviceToHost, c->stream));
    AUD_HIP(c, hipStreamSynchronize(c->stream));
    return AUD_OK;
}

}  // extern "C"

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
}  // namespace

extern "C" {

// ---- k-WTA ------------------------------------------------------------------------------

namespace {

// KWTA.Update(): fffb / nxx1 / chans derived values, float32 like the Go code (mat32.Pow goes through float64)
void fill_kwta_args(const aud_kwta_params* k, aud::KwtaArgs* a) {
    auto fffb = [](const aud_fffb_params& p) {
        aud::KwtaFffb f;
        f.on = p.on;
        f.gi = p.gi;
        f.ff = p.ff;
        f.fb = p.fb;
        f.fb_dt = 1.0f / p.fb_tau;
        f.max_vs_avg = p.max_vs_avg;
        f.ff0 = p.ff0;
        return f;
    };
    a->lay = fffb(k->lay_fffb);
    a->pool = fffb(k->pool_fffb);
    const aud_nxx1_params& x = k->xx1;
    a->gain = x.gain;
    a->nvar = x.nvar;
    a->interp_range = x.interp_range;
    a->gain_cor_range = x.gain_cor_range;
    a->gain_cor = x.gain_cor;
    a->sig_gain_nvar = x.sig_gain / x.nvar;
    a->sig_mult_eff = x.sig_mult * float(std::pow(double(x.gain * x.nvar), double(x.sig_mult_pow)));
    a->sig_val_at0 = 0.5f * a->sig_mult_eff;
    {  // XX1GainCor(InterpRange) - SigValAt0
        const float v = x.interp_range;
        const float fact = (x.gain_cor_range - (v / x.nvar)) / x.gain_cor_range;
        float y;
        if (fact < 0.f) {
            const float g = x.gain * v;
            y = g / (g + 1.f);
        } else {
            const float g = (x.gain * (1.f - x.gain_cor * fact)) * v;
            y = g / (g + 1.f);
        }
        a->interp_val = y - a->sig_val_at0;
    }
    a->gbar_e = k->gbar[0];
    a->gbar_l = k->gbar[1];
    a->gbar_i = k->gbar[2];
    a->erev_sub_thr_i = k->erev[2] - x.thr;
    a->erev_sub_thr_l = k->erev[1] - x.thr;
    a->thr_sub_erev_e = x.thr - k->erev[0];
    a->act_dt = 1.0f / k->act_tau;
    a->iters = k->iters;
    a->del_act_thr = k->del_act_thr;
}

}  // namespace

int aud_kwta_batch_dev(aud_ctx* c, const aud_kwta_params* k, const float* raw, float* act, int n_items, int d0,
                       int d1, int d2, int d3, int pool_level, int start_from_raw, float* pool_state,
                       int sum_order, int32_t* cycles, void* stream) {
    if (!c) return AUD_EINVAL;
    if (!k) return fail(c, AUD_EINVAL, "null parameters");
    if (n_items < 0 || d0 < 0 || d1 < 0 || d2 < 0 || d3 < 0) return fail(c, AUD_EINVAL, "bad shape");
    if ((pool_level != 0 && pool_level != 1) || (sum_order != 0 && sum_order != 1))
        return fail(c, AUD_EINVAL, "pool_level and sum_order are 0 or 1");
    if (k->iters < 0) return fail(c, AUD_EINVAL, "Iters < 0");
    if (!(k->act_tau != 0.f) || !(k->lay_fffb.fb_tau != 0.f) || !(k->pool_fffb.fb_tau != 0.f) || !(k->xx1.nvar != 0.f))
        return fail(c, AUD_EINVAL, "ActTau, FBTau and NVar must be non-zero");
    const int64_t n64 = int64_t(d0) * d1 * d2 * d3, lay64 = pool_level ? int64_t(d0) * d1 : 0;
    if (n_items == 0 || n64 == 0) return AUD_OK;
    if (!raw || !act) return fail(c, AUD_EINVAL, "null buffer");
    if (raw == act) return fail(c, AUD_EINVAL, "act must not alias raw (the reference keeps both tensors)");
    if (n64 > (1 << 20)) return fail(c, AUD_EINVAL, "tensor too large for one workgroup's LDS");
    // the zero-skipping list of the serial sum needs n more floats of LDS: used when they fit 64 KB with the rest
    bool compact = pool_level && sum_order == 0 && aud::kwta_lds_bytes(int(n64), int(lay64), true) <= 64u * 1024u;
    const size_t lds = aud::kwta_lds_bytes(int(n64), int(lay64), compact);
    if (lds > 160u * 1024u)
        return fail(c, AUD_EINVAL, "tensor too large for one workgroup's LDS: (32 + n + 4 pools) * 4 bytes must fit 160 KB");
    AUD_HIP(c, make_current(c));
    if (lds > 64u * 1024u) AUD_HIP(c, aud::kwta_prepare(unsigned(lds)));
    aud::KwtaArgs a;
    std::memset(&a, 0, sizeof(a));
    fill_kwta_args(k, &a);
    a.raw = raw;
    a.act = act;
    a.n_items = n_items;
    a.n = int(n64);
    a.lay_n = int(lay64);
    a.pl_n = pool_level ? d2 * d3 : 0;
    a.start_from_raw = start_from_raw ? 1 : 0;
    a.sum_order = sum_order;
    a.compact = compact ? 1 : 0;
    a.state = pool_level ? pool_state : nullptr;
    a.cycles = cycles;
    a.lds_bytes = unsigned(lds);
    AUD_HIP(c, aud::launch_kwta(a, static_cast<hipStream_t>(stream)));
    return AUD_OK;
}

int aud_kwta_batch_host(aud_ctx* c, const aud_kwta_params* k, const float* raw, float* act, int n_items, int d0,
                        int d1, int d2, int d3, int pool_level, int start_from_raw, float* pool_state,
                        int sum_order, int32_t* cycles) {
    if (!c) return AUD_EINVAL;
    if (n_items < 0 || d0 < 0 || d1 < 0 || d2 < 0 || d3 < 0) return fail(c, AUD_EINVAL, "bad shape");
    const size_t n = size_t(d0) * d1 * d2 * d3, total = size_t(n_items) * n;
    if (total == 0) return AUD_OK;
    if (!raw || !act) return fail(c, AUD_EINVAL, "null buffer");
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    const size_t n_state = (pool_level && pool_state) ? size_t(n_items) * d0 * d1 * 2 : 0;
    int rc;
    if ((rc = ensure_ws(c, 2, total * 4)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 3, total * 4)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 1, (n_state + size_t(n_items)) * 4 + 16)) != AUD_OK) return rc;
    float* d_raw = static_cast<float*>(c->ws[2]);
    float* d_act = static_cast<float*>(c->ws[3]);
    float* d_state = n_state ? static_cast<float*>(c->ws[1]) : nullptr;
    int32_t* d_cyc = reinterpret_cast<int32_t*>(static_cast<float*>(c->ws[1]) + n_state);
    AUD_HIP(c, hipMemcpyAsync(d_raw, raw, total * 4, hipMemcpyHostToDevice, c->stream));
    if (!start_from_raw) AUD_HIP(c, hipMemcpyAsync(d_act, act, total * 4, hipMemcpyHostToDevice, c->stream));
    if (n_state) AUD_HIP(c, hipMemcpyAsync(d_state, pool_state, n_state * 4, hipMemcpyHostToDevice, c->stream));
    rc = aud_kwta_batch_dev(c, k, d_raw, d_act, n_items, d0, d1, d2, d3, pool_level, start_from_raw, d_state,
                            sum_order, d_cyc, c->stream);
    if (rc != AUD_OK) {
        (void)hipStreamSynchronize(c->stream);
        return rc;
    }
    AUD_HIP(c, hipMemcpyAsync(act, d_act, total * 4, hipMemcpyDeviceToHost, c->stream));
    if (n_state) AUD_HIP(c, hipMemcpyAsync(pool_state, d_state, n_state * 4, hipMemcpyDeviceToHost, c->stream));
    if (cycles) AUD_HIP(c, hipMemcpyAsync(cycles, d_cyc, size_t(n_items) * 4, hipMemcpyDeviceToHost, c->stream));
    AUD_HIP(c, hipStreamSynchronize(c->stream));
    return AUD_OK;
}

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

int aud_gather_create(aud_ctx* c, int n_ranks, int rank, int64_t slab_floats, float** recv, char handle[64]) {
    if (!c || !recv || !handle || n_ranks < 1 || rank < 0 || rank >= n_ranks || slab_floats < 1) return AUD_EINVAL;
    if (c->gather.recv) return fail(c, AUD_EINVAL, "gather buffer already created");
    AUD_HIP(c, make_current(c));
    aud_ctx::Gather& g = c->gather;
    AUD_HIP(c, hipMalloc(reinterpret_cast<void**>(&g.recv), size_t(n_ranks) * size_t(slab_floats) * sizeof(float)));
    hipIpcMemHandle_t h;
    static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (hipIpcGetMemHandle(&h, g.recv) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(g.recv);
        g.recv = nullptr;
        return fail(c, AUD_EHIP, "hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
    }
    std::memcpy(handle, &h, 64);
    g.n_ranks = n_ranks;
    g.rank = rank;
    g.slab = slab_floats;
    g.peer.assign(size_t(n_ranks), nullptr);
    g.peer[size_t(rank)] = g.recv;
    g.streams.assign(size_t(n_ranks), nullptr);
    g.done.assign(size_t(n_ranks), nullptr);
    AUD_HIP(c, hipEventCreateWithFlags(&g.fork, hipEventDisableTiming));
    for (int p = 0; p < n_ranks; ++p) {
        if (p == rank) continue;
        AUD_HIP(c, hipStreamCreateWithFlags(&g.streams[size_t(p)], hipStreamNonBlocking));
        AUD_HIP(c, hipEventCreateWithFlags(&g.done[size_t(p)], hipEventDisableTiming));
    }
    *recv = g.recv;
    return AUD_OK;
}

int aud_gather_open_peer(aud_ctx* c, int peer, const char handle[64]) {
    if (!c || !handle) return AUD_EINVAL;
    aud_ctx::Gather& g = c->gather;
    if (!g.recv) return fail(c, AUD_EINVAL, "aud_gather_create has not been called");
    if (peer < 0 || peer >= g.n_ranks || peer == g.rank) return fail(c, AUD_EINVAL, "peer must be another rank of the gather");
    if (g.peer[size_t(peer)]) return fail(c, AUD_EINVAL, "peer already opened");
    AUD_HIP(c, make_current(c));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, 64);
    void* p = nullptr;
    AUD_HIP(c, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    g.peer[size_t(peer)] = static_cast<float*>(p);
    return AUD_OK;
}

int aud_allgather_direct_dev(aud_ctx* c, const float* send, int64_t count, void* stream) {
    if (!c || count < 0) return AUD_EINVAL;
    aud_ctx::Gather& g = c->gather;
    if (!g.recv) return fail(c, AUD_EINVAL, "aud_gather_create has not been called");
    if (count > g.slab) return fail(c, AUD_EINVAL, "count exceeds the slab the gather was created for");
    for (int p = 0; p < g.n_ranks; ++p)
        if (!g.peer[size_t(p)]) return fail(c, AUD_EINVAL, "a peer's receive buffer has not been opened");
    if (count == 0) return AUD_OK;
    if (!send) return fail(c, AUD_EINVAL, "null buffer");
    AUD_HIP(c, make_current(c));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t bytes = size_t(count) * sizeof(float), slot = size_t(g.rank) * size_t(g.slab);
    // own slot on the caller's stream; one push per peer, each on its own stream (its own xGMI link), forked from and
    // joined back into the caller's stream
    AUD_HIP(c, hipMemcpyAsync(g.recv + slot, send, bytes, hipMemcpyDeviceToDevice, st));
    if (g.n_ranks > 1) AUD_HIP(c, hipEventRecord(g.fork, st));
    for (int p = 0; p < g.n_ranks; ++p) {
        if (p == g.rank) continue;
        hipStream_t sp = g.streams[size_t(p)];
        AUD_HIP(c, hipStreamWaitEvent(sp, g.fork, 0));
        AUD_HIP(c, hipMemcpyAsync(g.peer[size_t(p)] + slot, send, bytes, hipMemcpyDeviceToDevice, sp));
        AUD_HIP(c, hipEventRecord(g.done[size_t(p)], sp));
        AUD_HIP(c, hipStreamWaitEvent(st, g.done[size_t(p)], 0));
    }
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
    }
    if (g.fork) (void)hipEventDestroy(g.fork);
    (void)hipFree(g.recv);
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
