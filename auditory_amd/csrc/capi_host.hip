// C ABI, host-staged entry points: the reference's per-step and per-segment calls on caller-owned host buffers, staged
// through the context's workspaces and stream (HostCallGuard).  See include/auditory_hip.h.
#include "capi_internal.h"

using namespace audc;

extern "C" {

namespace {

int check_items(aud_ctx* c, const aud_item* items, int n_items, int64_t sig_total) {
    for (int i = 0; i < n_items; ++i)
        if (items[i].sig_off < 0 || items[i].sig_len < 0 || items[i].sig_stride < 0 || item_last(items[i]) >= sig_total)
            return fail(c, AUD_EINVAL, "item outside the signal buffer");
    return AUD_OK;
}

// the frame loop on a signal that is already on the device (d_sig; the caller holds the HostCallGuard): items up, one
// launch, float32 results back through pinned staging, widened into the caller's float64 tensors
int melspec_host_run(aud_plan* p, const void* d_sig, int sig_dtype, const aud_item* items, int n_items, double* mel,
                     double* power, double* log_power) {
    aud_ctx* c = p->ctx;
    const int nf = p->d.mel.n_filters, T = p->d.segment_steps, H = p->H;
    const size_t n_mel = size_t(n_items) * nf * T, n_pow = size_t(n_items) * H * T;
    const size_t item_bytes = size_t(n_items) * sizeof(aud_item);
    const bool smooth = p->d.dft.prev_smooth != 0.0;  // the scan needs a device power buffer
    const bool want_p = power != nullptr || smooth, want_lp = log_power != nullptr;
    const size_t out_floats = n_mel + (want_p ? n_pow : 0) + (want_lp ? n_pow : 0);
    int rc;
    if ((rc = ensure_ws(c, 1, item_bytes)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 2, out_floats * 4)) != AUD_OK) return rc;
    float* d_mel = static_cast<float*>(c->ws[2]);
    float* d_pow = want_p ? d_mel + n_mel : nullptr;
    float* d_lp = want_lp ? d_mel + n_mel + (want_p ? n_pow : 0) : nullptr;
    AUD_HIP(c, hipMemcpyAsync(c->ws[1], items, item_bytes, hipMemcpyHostToDevice, c->stream));
    rc = aud_melspec_batch_dev(p, d_sig, sig_dtype, static_cast<const aud_item*>(c->ws[1]), n_items, d_mel, d_pow, d_lp, c->stream);
    if (rc != AUD_OK) return rc;
    const WidenPart parts[3] = {{mel, n_mel}, {power, want_p ? n_pow : 0}, {log_power, want_lp ? n_pow : 0}};
    if (all_parts_pinned(c, parts, 3)) return store_widened(c, d_mel, parts, 3);  // aud_host_alloc tensors: the device writes them
    return fetch_widened(c, d_mel, parts, 3);
}

// SndEnv.ProcessSegment with Mel.MFCC on, the same way
int melspec_mfcc_host_run(aud_plan* p, const void* d_sig, int sig_dtype, const aud_item* items, int n_items, double* mel,
                          double* power, double* log_power, double* mfcc, double* deltas, double* delta_deltas, double* energy) {
    aud_ctx* c = p->ctx;
    const int nf = p->d.mel.n_filters, T = p->d.segment_steps, H = p->H, nc = p->d.mfcc_coefs;
    const size_t n_mel = size_t(n_items) * nf * T, n_pow = size_t(n_items) * H * T;
    const size_t n_cc = size_t(n_items) * nc * T, n_en = size_t(n_items) * T;
    // device layout: mel | power | log_power | mfcc | deltas | delta_deltas | energy
    const size_t total = n_mel + 2 * n_pow + 3 * n_cc + n_en;
    const size_t item_bytes = size_t(n_items) * sizeof(aud_item);
    int rc;
    if ((rc = ensure_ws(c, 1, item_bytes)) != AUD_OK) return rc;
    if ((rc = ensure_ws(c, 2, total * 4)) != AUD_OK) return rc;
    float* d_mel = static_cast<float*>(c->ws[2]);
    float* d_pow = d_mel + n_mel;
    float* d_lp = d_pow + n_pow;
    float* d_cc = d_lp + n_pow;
    float* d_dl = d_cc + n_cc;
    float* d_ddl = d_dl + n_cc;
    float* d_en = d_ddl + n_cc;
    AUD_HIP(c, hipMemcpyAsync(c->ws[1], items, item_bytes, hipMemcpyHostToDevice, c->stream));
    const aud_item* d_items = static_cast<const aud_item*>(c->ws[1]);
    int64_t ws_bytes = 0;
    (void)aud_segment_workspace_bytes(p, n_items, &ws_bytes);
    if ((rc = ensure_ws(c, 3, size_t(ws_bytes) + 16)) != AUD_OK) return rc;
    rc = aud_segment_batch_dev(p, d_sig, sig_dtype, d_items, n_items, d_mel, d_pow, d_lp, d_cc, deltas ? d_dl : nullptr,
                               delta_deltas ? d_ddl : nullptr, d_en, c->ws[3], ws_bytes, c->stream);
    if (rc != AUD_OK) return rc;
    const WidenPart parts[7] = {{mel, n_mel}, {power, n_pow}, {log_power, n_pow}, {mfcc, n_cc}, {deltas, n_cc},
                                {delta_deltas, n_cc}, {energy, n_en}};
    if (all_parts_pinned(c, parts, 7)) return store_widened(c, d_mel, parts, 7);
    return fetch_widened(c, d_mel, parts, 7);
}

size_t sample_bytes(int dtype) { return dtype == AUD_F64 ? 8 : dtype == AUD_F32 ? 4 : 2; }

}  // namespace

int aud_melspec_batch_host(aud_plan* p, const double* sig, int64_t sig_total, const aud_item* items,
                           int n_items, double* mel, double* power, double* log_power) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (n_items < 0 || sig_total < 0 || (n_items > 0 && (!sig || !items || !mel)))
        return fail(c, AUD_EINVAL, "null buffer");
    if (n_items == 0) return AUD_OK;
    int rc = check_items(c, items, n_items, sig_total);
    if (rc != AUD_OK) return rc;
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    const size_t sig_bytes = size_t(sig_total) * 8;
    if ((rc = ensure_ws(c, 0, sig_bytes + 16)) != AUD_OK) return rc;
    AUD_HIP(c, hipMemcpyAsync(c->ws[0], sig, sig_bytes, hipMemcpyHostToDevice, c->stream));
    return melspec_host_run(p, c->ws[0], AUD_F64, items, n_items, mel, power, log_power);
}

int aud_host_alloc(aud_ctx* c, int64_t bytes, void** ptr) {
    if (!c || !ptr || bytes <= 0) return AUD_EINVAL;
    *ptr = nullptr;
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    void* p = nullptr;
    AUD_HIP(c, hipHostMalloc(&p, size_t(bytes), hipHostMallocDefault));
    c->host_blocks.push_back({static_cast<unsigned char*>(p), size_t(bytes), static_cast<unsigned char*>(p), false});
    *ptr = p;
    return AUD_OK;
}

int aud_host_register(aud_ctx* c, void* ptr, int64_t bytes) {
    if (!c || !ptr || bytes <= 0) return AUD_EINVAL;
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    {
        const unsigned char* lo = static_cast<const unsigned char*>(ptr);
        for (const auto& b : c->host_blocks)
            if (lo < b.p + b.bytes && b.p < lo + bytes)
                return fail(c, AUD_EINVAL, "aud_host_register: the range overlaps a block the context already holds");
    }
    AUD_HIP(c, hipHostRegister(ptr, size_t(bytes), hipHostRegisterPortable | hipHostRegisterMapped));
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess || !dev) {
        (void)hipGetLastError();
        (void)hipHostUnregister(ptr);
        return fail(c, AUD_EHIP, "aud_host_register: no device address for the range");
    }
    c->host_blocks.push_back({static_cast<unsigned char*>(ptr), size_t(bytes), static_cast<unsigned char*>(dev), true});
    return AUD_OK;
}

int aud_host_unregister(aud_ctx* c, void* ptr) {
    if (!c || !ptr) return AUD_EINVAL;
    (void)hipSetDevice(c->device);
    HostCallGuard guard(c);  // (drains the stream: no kernel still writes the range)
    for (size_t i = 0; i < c->host_blocks.size(); ++i)
        if (c->host_blocks[i].p == ptr && c->host_blocks[i].registered) {
            c->host_blocks.erase(c->host_blocks.begin() + long(i));
            (void)hipHostUnregister(ptr);
            return AUD_OK;
        }
    return fail(c, AUD_EINVAL, "aud_host_unregister: not a range of aud_host_register");
}

int aud_host_free(aud_ctx* c, void* ptr) {
    if (!c || !ptr) return AUD_EINVAL;
    (void)hipSetDevice(c->device);
    HostCallGuard guard(c);  // (drains the stream: no kernel still writes the block)
    for (size_t i = 0; i < c->host_blocks.size(); ++i)
        if (c->host_blocks[i].p == ptr && !c->host_blocks[i].registered) {
            c->host_blocks.erase(c->host_blocks.begin() + long(i));
            (void)hipHostFree(ptr);
            return AUD_OK;
        }
    return fail(c, AUD_EINVAL, "aud_host_free: not a block of aud_host_alloc");
}

int aud_signal_upload(aud_ctx* c, const void* samples, int sample_dtype, int64_t n_samples, aud_signal** out) {
    if (!c || !out) return AUD_EINVAL;
    *out = nullptr;
    if (sample_dtype != AUD_F64 && sample_dtype != AUD_F32 && sample_dtype != AUD_I16) return fail(c, AUD_EINVAL, "bad sample_dtype");
    if (n_samples < 0 || (n_samples > 0 && !samples)) return fail(c, AUD_EINVAL, "null buffer");
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    aud_signal* s = new (std::nothrow) aud_signal();
    if (!s) return AUD_ENOMEM;
    s->ctx = c;
    s->dtype = sample_dtype;
    s->n = n_samples;
    const size_t bytes = size_t(n_samples) * sample_bytes(sample_dtype);
    hipError_t e = hipMalloc(&s->d, bytes + 16);
    if (e == hipSuccess && bytes) e = hipMemcpyAsync(s->d, samples, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        if (s->d) (void)hipFree(s->d);
        delete s;
        return hip_fail(c, e, "aud_signal_upload");
    }
    {
        SignalRegistry& reg = SignalRegistry::get();
        std::lock_guard<std::mutex> lk(reg.m);
        reg.live.push_back(s);
    }
    *out = s;
    return AUD_OK;
}

int aud_signal_destroy(aud_signal* s) {
    if (!s) return AUD_EINVAL;
    aud_ctx* c = nullptr;
    {
        SignalRegistry& reg = SignalRegistry::get();
        std::lock_guard<std::mutex> lk(reg.m);
        auto it = std::find(reg.live.begin(), reg.live.end(), s);
        if (it == reg.live.end()) return AUD_EINVAL;  // not a live handle (destroyed twice?)
        reg.live.erase(it);
        c = s->ctx;  // null: the context was shut down first and took the device memory with it
    }
    if (c) {
        (void)hipSetDevice(c->device);
        HostCallGuard guard(c);  // (drains the context's stream: no call still reads the buffer)
        if (s->d) (void)hipFree(s->d);
    }
    delete s;
    return AUD_OK;
}

int64_t aud_signal_len(const aud_signal* s) { return s ? s->n : -1; }

namespace {

constexpr size_t kSyncBlock = 4096;  // compare / upload grain of a synced signal

// a handle of THIS context that is still alive (or null: to be created)
int check_live_signal(aud_ctx* c, aud_signal* s) {
    if (!s) return AUD_OK;
    SignalRegistry& reg = SignalRegistry::get();
    std::lock_guard<std::mutex> lk(reg.m);
    if (std::find(reg.live.begin(), reg.live.end(), s) == reg.live.end() || s->ctx != c)
        return fail(c, AUD_EINVAL, "signal of another (or a shut-down) context");
    return AUD_OK;
}

// The body of aud_signal_sync; the caller holds c->host_mutex.  `need` (may be null: every block) marks the kSyncBlock-byte
// blocks the coming call reads: only those are compared with the shadow, and what is uploaded is the span from the first to the
// last of them that differs.  Blocks outside stay as they are on BOTH sides (device copy == shadow always holds), so a later call
// that needs them finds their edits then.  Unchanged tensor: not a single runtime call.
int signal_sync_locked(aud_ctx* c, aud_signal** sig, const void* samples, int sample_dtype, int64_t n_samples,
                       const std::vector<unsigned char>* need, int64_t* uploaded_bytes) {
    aud_signal* s = *sig;
    if (!s) {
        s = new (std::nothrow) aud_signal();
        if (!s) return AUD_ENOMEM;
        s->ctx = c;
        SignalRegistry& reg = SignalRegistry::get();
        std::lock_guard<std::mutex> lk(reg.m);
        reg.live.push_back(s);
        *sig = s;
    }
    const unsigned char* src = static_cast<const unsigned char*>(samples);
    const size_t bytes = size_t(n_samples) * sample_bytes(sample_dtype);
    size_t lo = 0, hi = bytes;  // the span to upload
    const bool same_shape = s->shadow_ok && s->dtype == sample_dtype && s->n == n_samples;
    if (same_shape) {  // byte for byte against what the device holds
        lo = bytes;
        hi = 0;
        for (size_t b = 0, k = 0; b < bytes; b += kSyncBlock, ++k) {
            if (need && !(*need)[k]) continue;
            const size_t len = std::min(kSyncBlock, bytes - b);
            if (std::memcmp(src + b, s->shadow + b, len) != 0) {
                if (lo == bytes) lo = b;
                hi = b + len;
            }
        }
        if (lo >= hi) return AUD_OK;  // equal where it matters: the resident copy IS the caller's tensor there
    } else {
        s->shadow_ok = false;
        AUD_HIP(c, make_current(c));
        if (s->cap < bytes + 16) {
            if (c->stream) AUD_HIP(c, hipStreamSynchronize(c->stream));
            if (s->d) AUD_HIP(c, hipFree(s->d));
            s->d = nullptr;
            s->cap = 0;
            AUD_HIP(c, hipMalloc(&s->d, bytes + 16));
            s->cap = bytes + 16;
        }
        if (s->shadow_cap < bytes) {
            std::free(s->shadow);
            s->shadow = static_cast<unsigned char*>(std::malloc(bytes ? bytes : 1));
            s->shadow_cap = s->shadow ? bytes : 0;
            if (!s->shadow) return fail(c, AUD_ENOMEM, "aud_signal_sync: no memory for the shadow");
        }
        s->dtype = sample_dtype;
        s->n = n_samples;
    }
    s->shadow_ok = false;
    if (hi > lo) {  // (drained before returning: the caller may touch `samples` again, and the next call reads the new copy)
        AUD_HIP(c, make_current(c));
        AUD_HIP(c, hipMemcpyAsync(static_cast<unsigned char*>(s->d) + lo, src + lo, hi - lo, hipMemcpyHostToDevice, c->stream));
        std::memcpy(s->shadow + lo, src + lo, hi - lo);
        AUD_HIP(c, hipStreamSynchronize(c->stream));
    }
    s->shadow_ok = true;
    if (uploaded_bytes) *uploaded_bytes = int64_t(hi - lo);
    return AUD_OK;
}

// the blocks of a float64 signal the frames of `items` read under plan p: frame s of an item covers stream positions
// [start0 + S (s - border), + N) clipped to [0, sig_len) (sndenv.go:438-478), i.e. buffer elements sig_off + pos * stride
void mark_needed_blocks(const aud_plan* p, const aud_item* items, int n_items, size_t elem_bytes, size_t total_bytes,
                        std::vector<unsigned char>* need) {
    need->assign((total_bytes + kSyncBlock - 1) / kSyncBlock, 0);
    const int64_t S = p->d.step_samples, N = p->d.win_samples, T = p->d.segment_steps, border = p->d.border_steps;
    for (int i = 0; i < n_items; ++i) {
        const aud_item& it = items[i];
        const int64_t stride = it.sig_stride > 1 ? it.sig_stride : 1;
        int64_t first = int64_t(it.start0) - S * border, last = int64_t(it.start0) + S * (T - 1 - border) + N;  // [first, last)
        first = std::max<int64_t>(first, 0);
        last = std::min<int64_t>(last, it.sig_len);
        if (last <= first) continue;
        const size_t b0 = size_t(it.sig_off + first * stride) * elem_bytes / kSyncBlock;
        const size_t b1 = (size_t(it.sig_off + (last - 1) * stride) * elem_bytes + elem_bytes - 1) / kSyncBlock;
        for (size_t b = b0; b <= b1 && b < need->size(); ++b) (*need)[b] = 1;
    }
}

}  // namespace

int aud_signal_sync(aud_ctx* c, aud_signal** sig, const void* samples, int sample_dtype, int64_t n_samples,
                    int64_t* uploaded_bytes) {
    if (uploaded_bytes) *uploaded_bytes = 0;
    if (!c || !sig) return AUD_EINVAL;
    if (sample_dtype != AUD_F64 && sample_dtype != AUD_F32 && sample_dtype != AUD_I16) return fail(c, AUD_EINVAL, "bad sample_dtype");
    if (n_samples < 0 || (n_samples > 0 && !samples)) return fail(c, AUD_EINVAL, "null buffer");
    int rc = check_live_signal(c, *sig);
    if (rc != AUD_OK) return rc;
    // The compare needs the context's lock (the handle's shadow is shared state) but nothing of the device
    std::lock_guard<std::mutex> lock(c->host_mutex);
    return signal_sync_locked(c, sig, samples, sample_dtype, n_samples, nullptr, uploaded_bytes);
}

int aud_melspec_batch_live(aud_plan* p, aud_signal** sig, const double* samples, int64_t n_samples, const aud_item* items,
                           int n_items, double* mel, double* power, double* log_power, int64_t* uploaded_bytes) {
    if (uploaded_bytes) *uploaded_bytes = 0;
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (!sig || n_samples < 0 || n_items < 0 || (n_items > 0 && (!samples || !items || !mel))) return fail(c, AUD_EINVAL, "null buffer");
    if (n_items == 0) return AUD_OK;
    int rc = check_items(c, items, n_items, n_samples);
    if (rc == AUD_OK) rc = check_live_signal(c, *sig);
    if (rc != AUD_OK) return rc;
    std::vector<unsigned char> need;
    mark_needed_blocks(p, items, n_items, sizeof(double), size_t(n_samples) * sizeof(double), &need);
    HostCallGuard guard(c);
    if ((rc = signal_sync_locked(c, sig, samples, AUD_F64, n_samples, &need, uploaded_bytes)) != AUD_OK) return rc;
    AUD_HIP(c, make_current(c));
    return melspec_host_run(p, (*sig)->d, AUD_F64, items, n_items, mel, power, log_power);
}

int aud_melspec_mfcc_batch_live(aud_plan* p, aud_signal** sig, const double* samples, int64_t n_samples, const aud_item* items,
                                int n_items, double* mel, double* power, double* log_power, double* mfcc, double* deltas,
                                double* delta_deltas, double* energy, int64_t* uploaded_bytes) {
    if (uploaded_bytes) *uploaded_bytes = 0;
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (p->d.mfcc_coefs <= 0) return fail(c, AUD_EINVAL, "plan was created without mfcc_coefs");
    if (!p->d.dft.comp_log_pow) return fail(c, AUD_EINVAL, "the MFCC tail reads LogPowerSegment: needs CompLogPow");
    if (!sig || n_samples < 0 || n_items < 0 || (n_items > 0 && (!samples || !items || !mel || !mfcc)))
        return fail(c, AUD_EINVAL, "null buffer");
    if (delta_deltas && !deltas) return fail(c, AUD_EINVAL, "delta_deltas needs deltas");
    if (n_items == 0) return AUD_OK;
    int rc = check_items(c, items, n_items, n_samples);
    if (rc == AUD_OK) rc = check_live_signal(c, *sig);
    if (rc != AUD_OK) return rc;
    std::vector<unsigned char> need;
    mark_needed_blocks(p, items, n_items, sizeof(double), size_t(n_samples) * sizeof(double), &need);
    HostCallGuard guard(c);
    if ((rc = signal_sync_locked(c, sig, samples, AUD_F64, n_samples, &need, uploaded_bytes)) != AUD_OK) return rc;
    AUD_HIP(c, make_current(c));
    return melspec_mfcc_host_run(p, (*sig)->d, AUD_F64, items, n_items, mel, power, log_power, mfcc, deltas, delta_deltas, energy);
}

int aud_melspec_batch_sig(aud_plan* p, const aud_signal* s, const aud_item* items, int n_items, double* mel, double* power,
                          double* log_power) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (!s || s->ctx != c) return fail(c, AUD_EINVAL, "signal of another (or a shut-down) context, or null");
    if (n_items < 0 || (n_items > 0 && (!items || !mel))) return fail(c, AUD_EINVAL, "null buffer");
    if (n_items == 0) return AUD_OK;
    int rc = check_items(c, items, n_items, s->n);
    if (rc != AUD_OK) return rc;
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    return melspec_host_run(p, s->d, s->dtype, items, n_items, mel, power, log_power);
}

int aud_melspec_mfcc_batch_sig(aud_plan* p, const aud_signal* s, const aud_item* items, int n_items, double* mel, double* power,
                               double* log_power, double* mfcc, double* deltas, double* delta_deltas, double* energy) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (!s || s->ctx != c) return fail(c, AUD_EINVAL, "signal of another context (or null)");
    if (p->d.mfcc_coefs <= 0) return fail(c, AUD_EINVAL, "plan was created without mfcc_coefs");
    if (!p->d.dft.comp_log_pow) return fail(c, AUD_EINVAL, "the MFCC tail reads LogPowerSegment: needs CompLogPow");
    if (n_items < 0 || (n_items > 0 && (!items || !mel || !mfcc))) return fail(c, AUD_EINVAL, "null buffer");
    if (delta_deltas && !deltas) return fail(c, AUD_EINVAL, "delta_deltas needs deltas");
    if (n_items == 0) return AUD_OK;
    int rc = check_items(c, items, n_items, s->n);
    if (rc != AUD_OK) return rc;
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    return melspec_mfcc_host_run(p, s->d, s->dtype, items, n_items, mel, power, log_power, mfcc, deltas, delta_deltas, energy);
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
    int rc = check_items(c, items, n_items, sig_total);
    if (rc != AUD_OK) return rc;
    AUD_HIP(c, make_current(c));
    HostCallGuard guard(c);
    const size_t sig_bytes = size_t(sig_total) * 8;
    if ((rc = ensure_ws(c, 0, sig_bytes + 16)) != AUD_OK) return rc;
    AUD_HIP(c, hipMemcpyAsync(c->ws[0], sig, sig_bytes, hipMemcpyHostToDevice, c->stream));
    return melspec_mfcc_host_run(p, c->ws[0], AUD_F64, items, n_items, mel, power, log_power, mfcc, deltas, delta_deltas, energy);
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

