// C ABI of libauditory_hip.so: context / plan management and the device-resident batch entry points.
// See include/auditory_hip.h for the contract of every function.
#include "capi_internal.h"

namespace audc {

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
    a->bl_inplace = p->bl_inplace ? 1 : 0;
    a->bl_fix = (p->chirp_opt && p->d_bl_fix) ? p->d_bl_fix : nullptr;
    a->ip_nfac = plain_inplace(p) ? p->ip_nfac : 0;
    for (int i = 0; i < a->ip_nfac; ++i) a->ip_fac[i] = p->ip_fac[i];
    a->tw64 = p->direct ? p->d_tw64 : nullptr;
    a->xcd_remap = p->xcd_remap;
    a->stamps = reinterpret_cast<unsigned long long*>(p->stamps);
}

// the plan's frame -> power -> mel kernel (whatever family it selected), raw power, no smoothing
hipError_t launch_frames(const aud_plan* p, const aud::MelspecArgs& a, hipStream_t st) {
    if (p->use_wave && p->wave_kind) {
        if (p->lds_pad > 0 && p->wv.lds_bytes + unsigned(p->lds_pad) <= 64u * 1024u) {  // (plan option "lds_pad": fewer workgroups per CU)
            aud::WaveArgs e = p->wv;
            e.lds_bytes += unsigned(p->lds_pad);
            return aud::launch_melspec_wave(p->wave_kind, a, e, p->d.compute_dtype, st);
        }
        return aud::launch_melspec_wave(p->wave_kind, a, p->wv, p->d.compute_dtype, st);
    }
    if (p->direct) return aud::launch_melspec_direct(a, p->d.compute_dtype, st);
    if (a.bl_fix) {  // the chirp kernel transforms pairs of whole windows: length N whatever the any-N route's packing
        aud::MelspecArgs b = a;
        b.M = a.N;
        b.ratio = 1;
        b.F = 2;
        b.bl_chirp = p->d_fix_chirp;
        return aud::launch_melspec_chirp(b, st);
    }
    return aud::launch_melspec_generic(a, p->d.compute_dtype, st);
}

// smooth window lengths: the any-N kernel's in-place route runs this plan (melspec_generic.hip plain_fft_inplace)
bool plain_inplace(const aud_plan* p) { return p->bl_L == 0 && p->F_ip > 0 && (p->ip_opt != 0 || p->F_two < 1); }
// frames per workgroup of the any-N kernel on the route the plan's options select (Bluestein plans set theirs at creation)
void generic_route(aud_plan* p) {
    if (p->chirp_opt && p->d_bl_fix) p->F_generic = 2;   // the chirp kernel: a pair of frames per workgroup
    else if (p->bl_L) p->F_generic = p->F_bl;
    else p->F_generic = plain_inplace(p) ? p->F_ip : p->F_two;
}

const char* plan_family(const aud_plan* p) {
    if (p->direct) return "direct";
    if (!p->use_wave || !p->wave_kind) return (p->chirp_opt && p->d_bl_fix) ? "chirp2304" : "generic";  // (melspec_chirp.hip | any N)
    return p->wave_kind == 1 ? "w16x16" : p->wave_kind == 3 ? "w20x10" : "w64x16";
}

}  // namespace audc

using namespace audc;

namespace {

// as few Stockham stages as possible out of the radices the kernel has in registers
// (16, 8, 4, 2 | 25, 5 | 3), then whatever primes are left
// forward DFT of (re, im) in place, any length: decimation in time over the smallest prime factor (O(n sum of factors));
// long double throughout -- plan-time tables only
void fft_long_double(std::vector<long double>& re, std::vector<long double>& im) {
    const size_t n = re.size();
    if (n <= 1) return;
    size_t p = 2;
    while (p * p <= n && n % p != 0) ++p;
    if (n % p != 0) p = n;
    const long double pi = 3.14159265358979323846264338327950288L;
    const size_t m = n / p;
    std::vector<std::vector<long double>> sr(p, std::vector<long double>(m)), si(p, std::vector<long double>(m));
    for (size_t r = 0; r < p; ++r) {
        for (size_t j = 0; j < m; ++j) {
            sr[r][j] = re[j * p + r];
            si[r][j] = im[j * p + r];
        }
        if (m > 1) fft_long_double(sr[r], si[r]);
    }
    for (size_t k = 0; k < n; ++k) {  // X[k] = sum_r W_n^(r k) S_r[k mod m]
        long double ar = 0.0L, ai = 0.0L;
        for (size_t r = 0; r < p; ++r) {
            const long double ang = -2.0L * pi * (long double)((r * k) % n) / (long double)n;
            const long double c = cosl(ang), s = sinl(ang);
            const long double xr = sr[r][k % m], xi = si[r][k % m];
            ar += xr * c - xi * s;
            ai += xr * s + xi * c;
        }
        re[k] = ar;
        im[k] = ai;
    }
}

void factorize(int m, int* fac, int* nfac) {
    int n = 0;
    while (m % 16 == 0) { fac[n++] = 16; m /= 16; }
    while (m % 8 == 0) { fac[n++] = 8; m /= 8; }
    while (m % 4 == 0) { fac[n++] = 4; m /= 4; }
    while (m % 2 == 0) { fac[n++] = 2; m /= 2; }
    while (m % 25 == 0) { fac[n++] = 25; m /= 25; }
    while (m % 5 == 0) { fac[n++] = 5; m /= 5; }
    while (m % 9 == 0) { fac[n++] = 9; m /= 9; }
    while (m % 3 == 0) { fac[n++] = 3; m /= 3; }
    for (int p = 7; int64_t(p) * p <= m; p += 2)
        while (m % p == 0) { fac[n++] = p; m /= p; }
    if (m > 1) fac[n++] = m;
    *nfac = n;
}

// Bluestein's tables for a length-M DFT through transforms of length L >= 2 M - 1, in long double: chirp[n] = exp(-i pi n^2 / M)
// (n^2 reduced mod 2 M), bhat = FFT_L(conj chirp wrapped to length L) / L, twl[k] = exp(-2 pi i k / L)
void bluestein_tables(int M, int L, std::vector<double>& chirp, std::vector<double>& bhat, std::vector<double>& twl) {
    const long double pi = 3.14159265358979323846264338327950288L;
    const size_t Mz = size_t(M), Lz = size_t(L);
    std::vector<long double> wr(Mz), wi(Mz);
    for (int n = 0; n < M; ++n) {
        const long double ang = -pi * (long double)((int64_t(n) * n) % (2 * int64_t(M))) / (long double)M;
        wr[size_t(n)] = cosl(ang);
        wi[size_t(n)] = sinl(ang);
    }
    // b[m] = conj(w[|m|]) wrapped to length L, bhat = FFT_L(b) / L by a recursive mixed-radix FFT in long double (any L)
    std::vector<long double> br(Lz, 0.0L), bi(Lz, 0.0L);
    for (int m = 0; m < M; ++m) {
        br[size_t(m)] = wr[size_t(m)];
        bi[size_t(m)] = -wi[size_t(m)];
        if (m > 0) {
            br[size_t(L - m)] = wr[size_t(m)];
            bi[size_t(L - m)] = -wi[size_t(m)];
        }
    }
    fft_long_double(br, bi);
    chirp.assign(Mz * 2, 0.0);
    bhat.assign(Lz * 2, 0.0);
    twl.assign(Lz * 2, 0.0);
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
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    {  // resident signals of this context: their memory goes with it, the handles stay valid for aud_signal_destroy
        SignalRegistry& reg = SignalRegistry::get();
        std::lock_guard<std::mutex> lk(reg.m);
        for (aud_signal* s : reg.live)
            if (s->ctx == c) {
                if (s->d) (void)hipFree(s->d);
                s->d = nullptr;
                s->ctx = nullptr;
            }
    }
    for (int i = 0; i < 4; ++i)
        if (c->ws[i]) (void)hipFree(c->ws[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->pin) (void)hipHostFree(c->pin);
    for (auto& b : c->host_blocks) (void)(b.registered ? hipHostUnregister(b.p) : hipHostFree(b.p));
    for (auto& e : c->pin_ev)
        if (e) (void)hipEventDestroy(e);
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
    p->F_two = aud::melspec_generic_pick_F(p->M, d->compute_dtype);   // (0: its two buffers do not fit LDS)
    p->F_generic = p->F_two;

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
    // FFTs of a 2-3-5-smooth length L >= 2 M - 1 (melspec_generic.hip).  Tables in long double.
    {
        bool awkward = false;
        for (int i = 0; i < p->nfac; ++i) awkward = awkward || p->fac[i] > 25;
        const int L = awkward ? aud::melspec_generic_bluestein_L(p->M, d->compute_dtype) : 0;
        if (rc == AUD_OK && L > 0) {
            const int M = p->M;
            std::vector<double> chirp, bhat, twl;
            bluestein_tables(M, L, chirp, bhat, twl);
            rc = upload_real(c, &p->d_bl_chirp, chirp.data(), chirp.size(), d->compute_dtype);
            if (rc == AUD_OK) rc = upload_real(c, &p->d_bl_bhat, bhat.data(), bhat.size(), d->compute_dtype);
            if (rc == AUD_OK) rc = upload_real(c, &p->d_bl_tw, twl.data(), twl.size(), d->compute_dtype);
            const size_t lds = aud::melspec_generic_lds_bytes(L, 1, d->compute_dtype, true);
            if (rc == AUD_OK && lds > 64u * 1024u && aud::melspec_generic_prepare(lds) != hipSuccess) {
                (void)hipGetLastError();
                rc = fail(c, AUD_EHIP, "the runtime refused the LDS size of the Bluestein route");
            }
            if (rc == AUD_OK) {
                p->bl_L = L;
                factorize(L, p->bl_fac, &p->bl_nfac);
                p->bl_inplace = aud::melspec_generic_bluestein_inplace(L);  // one padded buffer, stages through registers
                // odd window lengths, float64 plans: two real frames per complex transform (float32 transforms keep one frame
                // each: separating a pair adds the partner's rounding floor, 4.2e-6 of the frame peak against 3e-6 measured)
                p->F_bl = (p->ratio == 1 && d->compute_dtype == AUD_F64) ? 2 : 1;
                p->F_generic = p->F_bl;
            }
        }
    }
    if (rc == AUD_OK && p->bl_L == 0) {
        // smooth lengths: the in-place route (one padded buffer, the workgroup's frames as one batched transform) where every
        // stage fits a thread's registers; the two-buffer route otherwise (radix 25 in one stage, the O(p) pass for 7 .. 23)
        size_t ip_lds = 0;
        p->F_ip = aud::melspec_generic_plain_inplace(p->M, H, nf, d->segment_steps, d->compute_dtype, 0, p->ip_fac, &p->ip_nfac, &ip_lds);
        const size_t two_lds = p->F_two > 0 ? aud::melspec_generic_lds_bytes(p->M, p->F_two, d->compute_dtype, false) : 0;
        if ((ip_lds > 64u * 1024u || two_lds > 64u * 1024u) && aud::melspec_generic_prepare(std::max(ip_lds, two_lds)) != hipSuccess) {
            (void)hipGetLastError();
            rc = fail(c, AUD_EHIP, "the runtime refused the LDS size of the any-N kernel");
        }
        if (rc == AUD_OK && p->F_ip < 1 && p->F_two < 1) {
            // no transform of this length fits a workgroup's LDS: the O(N H) kernel, which keeps only the spectrum there
            if (aud::melspec_direct_lds_bytes(H, nf, d->compute_dtype) > 160u * 1024u)
                rc = fail(c, AUD_EINVAL, "win_samples too large: the power spectrum of one frame must fit a workgroup's 160 KB of LDS");
            else if (aud::melspec_direct_prepare() != hipSuccess) {
                (void)hipGetLastError();
                rc = fail(c, AUD_EHIP, "the runtime refused the LDS size of the direct kernel");
            } else {
                std::vector<double> tw(size_t(N) * 2);
                const long double w = -2.0L * 3.14159265358979323846264338327950288L / (long double)N;
                for (int k = 0; k < N; ++k) {
                    tw[2 * size_t(k)] = double(cosl(w * k));
                    tw[2 * size_t(k) + 1] = double(sinl(w * k));
                }
                rc = upload(c, &p->d_tw64, tw.data(), tw.size() * sizeof(double));
                p->direct = rc == AUD_OK;
                p->F_two = 1;  // (one frame per workgroup: what F_generic, the fused tail's tiles and the launch use)
            }
        }
        generic_route(p);
    }
    bool chirp_pays = N > 512;
    if (!chirp_pays && p->F_ip < 1 && N >= 16) {
        // shorter windows (profiles/round6_rate_sweep.txt, "short awkward lengths"): the any-N kernel's Bluestein route costs 30-45 us per
        // 256 segments of 14 frames whatever its length -- one frame per workgroup on even windows (N = 254: 44.8 us), a pair on
        // odd ones (N = 331, L = 720: 30.5; N = 127, L = 256: 24.8) -- against this kernel's 27-29; its two-buffer route runs O(p)
        // passes for the primes 11 .. 23, over the whole window where N is odd (N = 276 = 4 x 3 x 23: 24.8 us -- kept; 275 = 11 x 25:
        // 32.3, 221 = 13 x 17: 40.0, 507 = 3 x 13^2: 73.7 -- all three taken over)
        int64_t odd_primes = 0;
        for (int i = 0; i < p->nfac; ++i)
            if (p->fac[i] > 9 && p->fac[i] != 16 && p->fac[i] != 25) odd_primes += p->fac[i];
        chirp_pays = p->bl_L ? (p->ratio == 2 || p->bl_L >= 512) : int64_t(p->ratio == 1 ? 2 : 1) * p->M * odd_primes > 4000;
    }
    if (rc == AUD_OK && !p->direct && p->F_ip < 1 && chirp_pays && aud::melspec_chirp_serves(N, d->compute_dtype)) {
        // The fixed-geometry chirp kernel (melspec_chirp.hip): every window 512 < N <= 1152 of a float64 plan that the smooth
        // in-place route does not run -- the reference's N = 1103 first of all -- as pairs of real frames through ONE chirp
        // convolution of length N on L = 2304, with tables of its own (whatever length the any-N route's Bluestein picked, if any)
        std::vector<double> chirp, bhat, twl, fix(size_t(aud::melspec_chirp_table_len()) * 2);
        bluestein_tables(N, 2304, chirp, bhat, twl);
        aud::melspec_chirp_tables(twl.data(), bhat.data(), fix.data());
        rc = upload_real(c, &p->d_bl_fix, fix.data(), fix.size(), d->compute_dtype);
        if (rc == AUD_OK) rc = upload_real(c, &p->d_fix_chirp, chirp.data(), chirp.size(), d->compute_dtype);
        generic_route(p);
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
    p->family = plan_family(p);
    if (rc == AUD_OK && p->wave_kind == 3) {  // the workgroup-per-item variant, where the item's mel matrix fits LDS beside the rest
        p->has_item = aud::melspec_item_finish(p->wave_kind, d->compute_dtype, p->wv, nf, d->segment_steps, &p->itm) &&
                      aud::melspec_item_prepare(p->wave_kind, d->compute_dtype, p->wv, &p->itm) == hipSuccess;
        (void)hipGetLastError();
    }
    if (rc == AUD_OK && d->n_gabor > 0) {
        // float32 copy of the taps for the LDS-staged gabor kernels, QUAD-INTERLEAVED [quads][SY][SX][4] (gabor_tile.h: the
        // filter pairs of one tap position are adjacent scalar-register pairs), zero rows past the last filter
        const size_t area = size_t(d->gabor.size_x) * d->gabor.size_y, quads = (size_t(d->n_gabor) + 3) / 4;
        std::vector<float> k32(quads * area * 4, 0.f);
        for (int gi = 0; gi < d->n_gabor; ++gi)
            for (size_t t = 0; t < area; ++t) k32[((size_t(gi) / 4) * area + t) * 4 + size_t(gi) % 4] = float(gabor_filters[size_t(gi) * area + t]);
        rc = upload(c, reinterpret_cast<void**>(&p->d_gabor32), k32.data(), k32.size() * 4);
    }
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
    if (p->d_bl_fix) (void)hipFree(p->d_bl_fix);
    if (p->d_tw64) (void)hipFree(p->d_tw64);
    if (p->d_fix_chirp) (void)hipFree(p->d_fix_chirp);
    if (p->d_filt) (void)hipFree(p->d_filt);
    if (p->d_bin_pts) (void)hipFree(p->d_bin_pts);
    if (p->d_gabor) (void)hipFree(p->d_gabor);
    if (p->d_gabor32) (void)hipFree(p->d_gabor32);
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
    const bool chirp = !wave && p->d_bl_fix && p->chirp_opt;  // the fixed-geometry kernel of L = 2304 runs this plan
    if (key == "lds_bytes") *value = wave ? int64_t(p->wv.lds_bytes) : chirp ? int64_t(aud::melspec_chirp_lds_bytes()) : 0;
    else if (key == "waves_per_wg") *value = wave ? p->wv.waves : 4;
    else if (key == "wgs_per_cu") *value = wave ? p->wv.wgs_per_cu : chirp ? int64_t((160u * 1024u) / aud::melspec_chirp_lds_bytes()) : 0;
    else if (key == "bluestein_L") *value = wave ? 0 : p->bl_L;
    else if (key == "bluestein_inplace") *value = !wave && p->bl_L && p->bl_inplace ? 1 : 0;
    else if (key == "chirp_kernel") *value = chirp ? 1 : 0;
    else if (key == "plain_inplace") *value = !wave && plain_inplace(p) ? 1 : 0;  // smooth length on the any-N kernel's in-place route
    else if (key == "generic_frames_per_wg") *value = p->F_generic;  // frames a workgroup of the any-N kernel transforms at once
    else if (key == "item_kernel") *value = wave && p->has_item ? 1 : 0;        // the workgroup-per-item variant exists for this plan
    else if (key == "item_waves") *value = wave && p->has_item ? p->itm.waves : 0;
    else if (key == "item_lds_bytes") *value = wave && p->has_item ? int64_t(p->itm.lds_bytes) : 0;
    else if (key == "item_wgs_per_cu") *value = wave && p->has_item ? p->itm.wgs_per_cu : 0;
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
    if (key == "item_kernel") {  // -1 (default): fused mel + gabor calls only; 0: never; 1: every call the variant can serve
        if (value < -1 || value > 1) return fail(c, AUD_EINVAL, "item_kernel: -1 (auto), 0 (off) or 1 (on)");
        p->item_opt = value;
        return AUD_OK;
    }
    // -1 (default): float64 plans take k_gabor (float64 taps, every multiply-add in float64: gabor.go:268-283 as written), float32
    // plans the LDS-staged kernel where the item fits; 0: the LDS-staged kernel (float32 taps and row sums: for a float64 plan an
    // explicit opt-in, ~1e-6 of the all-float64 sum and a sign that can differ where fSum ~ 0); 1: one thread per position
    if (key == "gabor_kernel") {
        if (value < -1 || value > 1) return fail(c, AUD_EINVAL, "gabor_kernel: -1 (auto), 0 (LDS-staged, float32 taps) or 1 (one thread per position)");
        p->gabor_opt = value;
        return AUD_OK;
    }
    // extra dynamic LDS per workgroup of the wave kernels (bytes, the total stays <= 64 KB): lowers the workgroups a CU holds --
    // and with them the registers the kernel's waves take -- so that a second kernel's waves find room beside them
    if (key == "lds_pad") {
        if (value < 0 || value > 64 * 1024) return fail(c, AUD_EINVAL, "lds_pad: 0 .. 65536 bytes");
        p->lds_pad = value;
        return AUD_OK;
    }
    // -1 / 1 (default): aud_segment_batch_dev lets the plan's mel kernel carry the MFCC tail wherever it can (w20x10, w16x16, the
    // any-N kernel); 0: always the two launches on the float32-STORED tensors (aud_melspec_batch_dev + aud_mfcc_batch_dev)
    if (key == "fused_tail") {
        if (value < -1 || value > 1) return fail(c, AUD_EINVAL, "fused_tail: -1 / 1 (wherever the kernel can) or 0 (never)");
        p->fused_tail_opt = value;
        return AUD_OK;
    }
    if (key == "plain_inplace") {  // 1 (default): smooth lengths run the any-N kernel's in-place route where it serves; 0: two buffers
        if (value != 0 && value != 1) return fail(c, AUD_EINVAL, "plain_inplace: 0 or 1");
        p->ip_opt = value;
        generic_route(p);
        return AUD_OK;
    }
    if (key == "plain_frames") {  // frames per workgroup of the in-place route: 0 = the plan's own choice, else 1 / 2 / 4 / 8 / 16
        if (p->bl_L != 0) return fail(c, AUD_EINVAL, "plain_frames: the plan runs the Bluestein route");
        int fac[aud::kMaxFactors], nfac = 0;
        size_t lds = 0;
        const int F = aud::melspec_generic_plain_inplace(p->M, p->H, p->d.mel.n_filters, p->d.segment_steps, p->d.compute_dtype,
                                                         value, fac, &nfac, &lds);
        if (value < 0 || F < 1) return fail(c, AUD_EINVAL, "plain_frames: the in-place route does not run this length with that many frames");
        if (lds > 64u * 1024u && aud::melspec_generic_prepare(lds) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, AUD_EHIP, "the runtime refused the LDS size of the any-N kernel");
        }
        p->F_ip = F;
        generic_route(p);
        return AUD_OK;
    }
    if (key == "chirp_kernel") {  // 1 (default): the fixed-geometry chirp kernel wherever it serves the plan; 0: the any-N route
        if (value != 0 && value != 1) return fail(c, AUD_EINVAL, "chirp_kernel: 0 or 1");
        p->chirp_opt = value;
        generic_route(p);
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
    if ((power || log_power) && int64_t(p->H) * int64_t(p->d.segment_steps) >= (int64_t(1) << 29))
        return fail(c, AUD_EINVAL, "an item's [H, T] spectrum tensor must stay below 2 GB (it sits behind a buffer descriptor)");
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
    if (p->item_opt == 1 && p->has_item && p->use_wave && !smooth) {  // the workgroup-per-item variant, no gabor phase
        aud::ItemArgs g = p->itm;
        g.nG = 0;
        AUD_HIP(c, aud::launch_melspec_item(p->wave_kind, a, p->wv, g, p->d.compute_dtype, static_cast<hipStream_t>(stream)));
        return AUD_OK;
    }
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

namespace {
// the plan's mel kernel can carry the tail's DCT and Energy sums itself (kernels.h MelspecArgs::mfcc_acc)
bool segment_fused(const aud_plan* p) {
    if (p->fused_tail_opt == 0 || p->d.dft.prev_smooth != 0.0 || p->d.mfcc_coefs < 1 ||
        aud::segment_finish_lds_bytes(p->d.mfcc_coefs, p->d.segment_steps, p->d.compute_dtype) > 64 * 1024)
        return false;
    if (p->use_wave && p->wave_kind) return p->wv.dct_off >= 0;
    // the any-N kernel (round 6): DCT and Energy sums from its unrounded values wherever its power buffer has the room
    if (p->chirp_opt && p->d_bl_fix) return aud::melspec_chirp_tail_fits(p->H, p->d.mel.n_filters);
    if (plain_inplace(p) || p->direct) return true;  // (their launches' LDS is sized with the tail's F x nf values)
    return aud::melspec_generic_tail_fits(p->M, p->F_generic, p->H, p->d.mel.n_filters, p->d.compute_dtype, p->bl_L, p->bl_inplace);
}
// tiles of an item the fused tail's per-tile Energy sums come in: wave tiles, or the any-N kernel's workgroups of F frames
int segment_tiles(const aud_plan* p) {
    const int fw = (p->use_wave && p->wave_kind) ? aud::melspec_wave_frames_per_wave(p->wave_kind) : p->F_generic;
    return (p->d.segment_steps + fw - 1) / fw;
}
size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
}  // namespace

int aud_segment_workspace_bytes(const aud_plan* p, int n_items, int64_t* bytes) {
    if (!p || !bytes || n_items < 0) return AUD_EINVAL;
    const size_t T = size_t(p->d.segment_steps), n = size_t(n_items);
    if (segment_fused(p)) {
        const size_t tiles = size_t(segment_tiles(p)), tsz = p->d.compute_dtype == AUD_F64 ? 8 : 4;
        *bytes = int64_t(align256(n * p->d.mfcc_coefs * T * tsz) + align256(n * tiles * T * tsz));
    } else {
        // a LogPowerSegment of its own when the caller keeps none, and -- PrevSmooth != 0: the scan runs on the stored
        // power tensor -- a PowerSegment too
        *bytes = int64_t(align256(n * p->H * T * 4) * (p->d.dft.prev_smooth != 0.0 ? 2 : 1));
    }
    return AUD_OK;
}

int aud_segment_batch_dev(aud_plan* p, const void* sig, int sig_dtype, const aud_item* items, int n_items, float* mel,
                          float* power, float* log_power, float* mfcc, float* deltas, float* delta_deltas, float* energy,
                          void* workspace, int64_t workspace_bytes, void* stream) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    if (p->d.mfcc_coefs <= 0 || !p->d_dct) return fail(c, AUD_EINVAL, "plan was created without mfcc_coefs");
    if (!p->d.dft.comp_log_pow) return fail(c, AUD_EINVAL, "the MFCC tail reads LogPowerSegment: needs CompLogPow");
    if (p->d.segment_steps > p->H)
        return fail(c, AUD_EINVAL, "Energy reads LogPowerSegment row s < SegmentSteps: needs SegmentSteps <= bins (Go panics)");
    if (delta_deltas && !deltas) return fail(c, AUD_EINVAL, "delta_deltas needs deltas");
    if (n_items < 0 || (n_items > 0 && (!sig || !items || !mel || !mfcc))) return fail(c, AUD_EINVAL, "null buffer");
    int64_t need = 0;
    (void)aud_segment_workspace_bytes(p, n_items, &need);
    if (n_items > 0 && (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15)))
        return fail(c, AUD_EINVAL, "workspace: 16-byte aligned, aud_segment_workspace_bytes() bytes");
    if (n_items == 0) return AUD_OK;
    if (!segment_fused(p)) {  // (w64x16, PrevSmooth, more than 13 coefficients on a wave kernel, option fused_tail = 0): the two launches of the parts
        float* lp = log_power ? log_power : static_cast<float*>(workspace);
        float* pw = power;
        if (!pw && p->d.dft.prev_smooth != 0.0)  // `power` stays optional: the scan's tensor comes out of the workspace
            pw = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + align256(size_t(n_items) * p->H * p->d.segment_steps * 4));
        int rc = aud_melspec_batch_dev(p, sig, sig_dtype, items, n_items, mel, pw, lp, stream);
        if (rc != AUD_OK) return rc;
        return aud_mfcc_batch_dev(p, items, n_items, mel, lp, mfcc, deltas, delta_deltas, energy, stream);
    }
    if (sig_dtype != AUD_F32 && sig_dtype != AUD_F64 && sig_dtype != AUD_I16) return fail(c, AUD_EINVAL, "bad sig_dtype");
    if (int64_t(n_items) * int64_t(p->d.segment_steps) > (int64_t(1) << 30))
        return fail(c, AUD_EINVAL, "n_items x segment_steps too large for one launch; split the batch");
    if ((power || log_power) && int64_t(p->H) * int64_t(p->d.segment_steps) >= (int64_t(1) << 29))
        return fail(c, AUD_EINVAL, "an item's [H, T] spectrum tensor must stay below 2 GB (it sits behind a buffer descriptor)");
    AUD_HIP(c, make_current(c));
    const size_t T = size_t(p->d.segment_steps), tsz = p->d.compute_dtype == AUD_F64 ? 8 : 4;
    aud::MelspecArgs a;
    fill_melspec_args(p, &a);
    a.sig = sig;
    a.sig_dtype = sig_dtype;
    a.items = items;
    a.n_items = n_items;
    a.mel = mel;
    a.power = power;
    a.log_power = log_power;
    a.mfcc_acc = workspace;
    a.energy_part = static_cast<unsigned char*>(workspace) + align256(size_t(n_items) * p->d.mfcc_coefs * T * tsz);
    a.n_coefs = p->d.mfcc_coefs;
    a.dct_rows = p->d_dct;
    AUD_HIP(c, launch_frames(p, a, static_cast<hipStream_t>(stream)));
    aud::SegmentFinishArgs f;
    std::memset(&f, 0, sizeof(f));
    f.n_items = n_items;
    f.T = p->d.segment_steps;
    f.n_coefs = p->d.mfcc_coefs;
    f.tiles = segment_tiles(p);
    f.mfcc_acc = a.mfcc_acc;
    f.energy_part = a.energy_part;
    f.mfcc = mfcc;
    f.deltas = deltas;
    f.delta_deltas = delta_deltas;
    f.energy = energy;
    AUD_HIP(c, aud::launch_segment_finish(f, p->d.compute_dtype, static_cast<hipStream_t>(stream)));
    return AUD_OK;
}

namespace {
// Convolve's iteration space and envelope for one call (gabor.go:226-262): fills everything of GaborArgs but the
// buffers.  AUD_OK with nT == 0 or nF == 0: nothing to do.
int gabor_geometry(aud_plan* p, int n_items, int rows, int cols, int out_rank, const int32_t* out_shape, int by_time,
                   aud::GaborArgs* out) {
    aud_ctx* c = p->ctx;
    if (p->d.n_gabor <= 0 || !p->d_gabor) return fail(c, AUD_EINVAL, "plan has no gabor filters");
    if (n_items < 0 || rows < 1 || cols < 1 || !out_shape) return fail(c, AUD_EINVAL, "bad shape");
    const aud_gabor_set& g = p->d.gabor;
    int32_t nT = 0, nF = 0, strides = 1;
    if (aud_gabor_iter_space(&g, rows, cols, out_rank, out_shape, &nT, &nF, &strides) != AUD_OK)
        return fail(c, AUD_EINVAL, "Convolve rejects this shape (gabor.go:226-229, :259-262)");
    aud::GaborArgs a;
    std::memset(&a, 0, sizeof(a));
    a.nT = nT;
    a.nF = nF;
    a.n_items = n_items;
    *out = a;
    if (n_items == 0 || nT == 0 || nF == 0) return AUD_OK;
    const int nG = p->d.n_gabor;
    // reads: the reference indexes melData by flat offset; past the end it panics
    const int64_t last_read = int64_t((nF - 1) * g.stride_y + g.size_y - 1) * cols +
                              int64_t(nT - 1) * g.stride_x + g.size_x - 1;
    if (last_read >= int64_t(rows) * cols)
        return fail(c, AUD_EINVAL, "gabor pools reach past the mel matrix (SURVEY Q10)");
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
    a.rows = rows;
    a.cols = cols;
    a.k = p->d_gabor;
    a.k32 = p->d_gabor32;
    a.mode = p->gabor_opt;
    a.nG = nG;
    a.SX = g.size_x;
    a.SY = g.size_y;
    a.stx = g.stride_x;
    a.sty = g.stride_y;
    a.gain = g.gain;
    a.rank = out_rank;
    a.by_time = by_time;
    a.t_max_strides = strides;
    *out = a;
    return AUD_OK;
}
}  // namespace

int aud_gabor_batch_dev(aud_plan* p, const float* mel, int n_items, int rows, int cols, int out_rank,
                        const int32_t* out_shape, int by_time, float* out, void* stream) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    aud::GaborArgs a;
    const int rc = gabor_geometry(p, n_items, rows, cols, out_rank, out_shape, by_time, &a);
    if (rc != AUD_OK) return rc;
    if (n_items == 0 || a.nT == 0 || a.nF == 0) return AUD_OK;
    if (!mel || !out) return fail(c, AUD_EINVAL, "null buffer");
    AUD_HIP(c, make_current(c));
    a.mel = mel;
    a.out = out;
    AUD_HIP(c, aud::launch_gabor(a, p->d.compute_dtype, static_cast<hipStream_t>(stream)));
    return AUD_OK;
}

int aud_process_batch_dev(aud_plan* p, const void* sig, int sig_dtype, const aud_item* items,
                          int n_items, float* mel, int pools_y, int pools_x, float* gabor,
                          void* stream) {
    if (!p) return AUD_EINVAL;
    aud_ctx* c = p->ctx;
    const int32_t shape[4] = {pools_y, pools_x, 2, p->d.n_gabor};
    // ONE launch where the plan has the workgroup-per-item kernel (N = 400): the item's mel matrix stays in LDS behind the
    // frame loop and Convolve runs on it there (melspec_w20.hip k_melspec_w20_item); otherwise the two launches
    // (option "item_kernel" = 1; the default is the two launches: at 256 items per launch the workgroup-per-item kernel keeps a
    // CU's vector ALUs less busy than the tile kernel does -- DESIGN.md 4.5 has the measurements)
    const bool fused = p->has_item && p->use_wave && p->item_opt == 1 && p->d.dft.prev_smooth == 0.0 && p->d_gabor32 &&
                       shape[3] == p->d.n_gabor && shape[2] == 2;
    if (fused) {
        aud::GaborArgs ga;
        int rc = gabor_geometry(p, n_items, p->d.mel.n_filters, p->d.segment_steps, 4, shape, 0, &ga);
        if (rc != AUD_OK) return rc;
        if (n_items < 0 || (n_items > 0 && (!sig || !items || !mel))) return fail(c, AUD_EINVAL, "null buffer");
        if (sig_dtype != AUD_F32 && sig_dtype != AUD_F64 && sig_dtype != AUD_I16) return fail(c, AUD_EINVAL, "bad sig_dtype");
        if (n_items == 0) return AUD_OK;
        if (ga.nT > 0 && ga.nF > 0 && !gabor) return fail(c, AUD_EINVAL, "null buffer");
        AUD_HIP(c, make_current(c));
        aud::MelspecArgs a;
        fill_melspec_args(p, &a);
        a.sig = sig;
        a.sig_dtype = sig_dtype;
        a.items = items;
        a.n_items = n_items;
        a.mel = mel;
        aud::ItemArgs g = p->itm;
        g.k32 = p->d_gabor32;
        g.nG = (ga.nT > 0 && ga.nF > 0) ? p->d.n_gabor : 0;  // an empty iteration space: Convolve writes nothing
        g.SX = ga.SX;
        g.SY = ga.SY;
        g.stx = ga.stx;
        g.sty = ga.sty;
        g.gain = ga.gain;
        g.rank = 4;
        g.d0 = ga.d0;
        g.d1 = ga.d1;
        g.d2 = ga.d2;
        g.d3 = ga.d3;
        g.nT = ga.nT;
        g.nF = ga.nF;
        g.out = gabor;
        AUD_HIP(c, aud::launch_melspec_item(p->wave_kind, a, p->wv, g, p->d.compute_dtype, static_cast<hipStream_t>(stream)));
        return AUD_OK;
    }
    {  // the Convolve shape is checked BEFORE anything is launched: a rejected call writes nothing, mel included
        aud::GaborArgs ga;
        const int rc = gabor_geometry(p, n_items, p->d.mel.n_filters, p->d.segment_steps, 4, shape, 0, &ga);
        if (rc != AUD_OK) return rc;
        if (n_items > 0 && ga.nT > 0 && ga.nF > 0 && !gabor) return fail(c, AUD_EINVAL, "null buffer");
    }
    int rc = aud_melspec_batch_dev(p, sig, sig_dtype, items, n_items, mel, nullptr, nullptr, stream);
    if (rc != AUD_OK) return rc;
    return aud_gabor_batch_dev(p, mel, n_items, p->d.mel.n_filters, p->d.segment_steps, 4, shape, 0,
                               gabor, stream);
}


}  // extern "C"

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


}  // extern "C"
