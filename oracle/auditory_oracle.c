/*
 * auditory_oracle.c -- CPU float64 restatement of the emer/auditory hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under auditory_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU stand-in.
 *
 * PARITY UNPINNED: the reference (emer/auditory v0.9.8, Go) ships no tests,
 * golden vectors or known-answer values, and cannot be built in this image
 * (no Go toolchain, module deps not vendored).  This restatement is pinned
 * instead by (i) first-principles KATs, (ii) an independent numpy/long-double
 * DFT cross-check (tests/test_oracle.py) -- see DESIGN.md "Oracle".
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).  Arithmetic is float64 throughout, as in the reference;
 * only the gabor output is rounded to float32 (etensor.Float32 store).
 *
 * The FFT itself lives in gonum v0.11.0 dsp/fourier (FFTPACK cfftf port), which
 * is not vendored in the reference; its published contract -- forward,
 * unnormalised X[k] = sum_j x[j] exp(-2 pi i j k / n), any n, float64 -- is
 * restated here with an own mixed-radix FFT (any correct float64 DFT agrees
 * to ~1e-13 relative).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define ORC_OK 0
#define ORC_EINVAL 1   /* shape rejected by the reference ("log + return") */
#define ORC_EPANIC 2   /* the Go code would panic (slice index out of range) */
#define ORC_ESHORT 3   /* SndToWindow: end beyond signal length (sndenv.go:458-460) */

/* ------------------------------------------------------------------ */
/* sound/sndenv.go:522-524  MSecToSamples                              */
/* ------------------------------------------------------------------ */
int orc_msec_to_samples(double ms, int rate) {
    /* Go math.Round = half away from zero = C round() */
    return (int)round(ms * 0.001 * (double)rate);
}

/* sound/sndenv.go:202-207: derived sample counts */
void orc_sound_params(double win_ms, double step_ms, double segment_ms, double stride_ms,
                      int border_steps, int sr, int* win_samples, int* step_samples,
                      int* segment_samples, int* stride_samples, int* segment_steps) {
    *win_samples = orc_msec_to_samples(win_ms, sr);
    *step_samples = orc_msec_to_samples(step_ms, sr);
    *segment_samples = orc_msec_to_samples(segment_ms, sr);
    int steps = (int)round(segment_ms / step_ms);
    *segment_steps = steps + 2 * border_steps;
    *stride_samples = orc_msec_to_samples(stride_ms, sr);
}

/* sound/sndenv.go:263-265: SegCnt */
int orc_seg_cnt(int signal_len, int segment_samples, int stride_samples, int channels) {
    int siglen = signal_len - segment_samples * channels;
    siglen = siglen / channels; /* Go int division truncates toward zero, like C */
    return siglen / stride_samples + 1;
}

/* sound/sound.go:130-141 GetFloatAtIdx: int PCM -> float64 */
double orc_pcm_to_float(int v, int bit_depth) {
    if (bit_depth == 32) return (double)v / (double)0x7FFFFFFF;
    if (bit_depth == 24) return (double)v / (double)0x7FFFFF;
    if (bit_depth == 16) return (double)v / (double)0x7FFF;
    if (bit_depth == 8) return (double)v / (double)0x7F;
    return 0;
}

/* ------------------------------------------------------------------ */
/* Forward unnormalised complex DFT, float64, any n                    */
/* (contract of gonum fourier.CmplxFFT.Coefficients; dft/dft.go:45-46) */
/* ------------------------------------------------------------------ */
typedef struct {
    int n;
    int nfac;
    int fac[64];
    double* tw; /* 2*n: W_n^k = exp(-2 pi i k / n) */
    double* scratch; /* 2*n */
} orc_fft_plan;

orc_fft_plan* orc_fft_plan_create(int n) {
    if (n <= 0) return NULL;
    orc_fft_plan* p = (orc_fft_plan*)calloc(1, sizeof(orc_fft_plan));
    p->n = n;
    int m = n;
    /* radix-4 first, then 2, 3, 5, then remaining primes */
    while (m % 4 == 0) { p->fac[p->nfac++] = 4; m /= 4; }
    while (m % 2 == 0) { p->fac[p->nfac++] = 2; m /= 2; }
    for (int f = 3; (long)f * f <= m; f += 2)
        while (m % f == 0) { p->fac[p->nfac++] = f; m /= f; }
    if (m > 1) p->fac[p->nfac++] = m;
    p->tw = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    p->scratch = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    for (int k = 0; k < n; k++) {
        long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
        p->tw[2 * k] = (double)cosl(a);
        p->tw[2 * k + 1] = (double)sinl(a);
    }
    return p;
}

void orc_fft_plan_destroy(orc_fft_plan* p) {
    if (!p) return;
    free(p->tw);
    free(p->scratch);
    free(p);
}

/* out[k] (k<n) = DFT of in[0], in[stride], ..., in[(n-1)*stride]; tws = N/n */
static void fft_rec(const orc_fft_plan* p, int depth, const double* in, double* out, int n,
                    int stride, int tws) {
    if (n == 1) {
        out[0] = in[0];
        out[1] = in[1];
        return;
    }
    const int r = p->fac[depth];
    const int m = n / r;
    for (int j = 0; j < r; j++)
        fft_rec(p, depth + 1, in + 2 * (size_t)j * stride, out + 2 * (size_t)j * m, m, stride * r,
                tws * r);
    const int N = p->n;
    const double* tw = p->tw;
    if (r == 2) {
        for (int k = 0; k < m; k++) {
            double ar = out[2 * k], ai = out[2 * k + 1];
            double br = out[2 * (k + m)], bi = out[2 * (k + m) + 1];
            double wr = tw[2 * (k * tws)], wi = tw[2 * (k * tws) + 1];
            double tr = br * wr - bi * wi, ti = br * wi + bi * wr;
            out[2 * k] = ar + tr;
            out[2 * k + 1] = ai + ti;
            out[2 * (k + m)] = ar - tr;
            out[2 * (k + m) + 1] = ai - ti;
        }
        return;
    }
    if (r == 4) {
        for (int k = 0; k < m; k++) {
            double y[4][2];
            y[0][0] = out[2 * k];
            y[0][1] = out[2 * k + 1];
            for (int j = 1; j < 4; j++) {
                double br = out[2 * (k + j * m)], bi = out[2 * (k + j * m) + 1];
                int ti = (int)(((long)j * k * tws) % N);
                double wr = tw[2 * ti], wi = tw[2 * ti + 1];
                y[j][0] = br * wr - bi * wi;
                y[j][1] = br * wi + bi * wr;
            }
            double s0r = y[0][0] + y[2][0], s0i = y[0][1] + y[2][1];
            double d0r = y[0][0] - y[2][0], d0i = y[0][1] - y[2][1];
            double s1r = y[1][0] + y[3][0], s1i = y[1][1] + y[3][1];
            double d1r = y[1][0] - y[3][0], d1i = y[1][1] - y[3][1];
            /* forward: W_4 = -i */
            out[2 * k] = s0r + s1r;
            out[2 * k + 1] = s0i + s1i;
            out[2 * (k + m)] = d0r + d1i;
            out[2 * (k + m) + 1] = d0i - d1r;
            out[2 * (k + 2 * m)] = s0r - s1r;
            out[2 * (k + 2 * m) + 1] = s0i - s1i;
            out[2 * (k + 3 * m)] = d0r - d1i;
            out[2 * (k + 3 * m) + 1] = d0i + d1r;
        }
        return;
    }
    /* generic radix r: X[k + q m] = sum_j W_n^{j k} W_r^{j q} Y_j[k] */
    double tstack[64];
    double* t = (r <= 32) ? tstack : (double*)malloc(sizeof(double) * 2 * (size_t)r);
    double ostack[64];
    double* o = (r <= 32) ? ostack : p->scratch;
    const int wr_stride = N / r;
    for (int k = 0; k < m; k++) {
        for (int j = 0; j < r; j++) {
            double br = out[2 * (k + j * m)], bi = out[2 * (k + j * m) + 1];
            int ti = (int)(((long)j * k * tws) % N);
            double wr = tw[2 * ti], wi = tw[2 * ti + 1];
            t[2 * j] = br * wr - bi * wi;
            t[2 * j + 1] = br * wi + bi * wr;
        }
        for (int q = 0; q < r; q++) {
            double sr = 0, si = 0;
            for (int j = 0; j < r; j++) {
                int ti = (int)(((long)j * q) % r) * wr_stride;
                double wr = tw[2 * ti], wi = tw[2 * ti + 1];
                sr += t[2 * j] * wr - t[2 * j + 1] * wi;
                si += t[2 * j] * wi + t[2 * j + 1] * wr;
            }
            o[2 * q] = sr;
            o[2 * q + 1] = si;
        }
        for (int q = 0; q < r; q++) {
            out[2 * (k + q * m)] = o[2 * q];
            out[2 * (k + q * m) + 1] = o[2 * q + 1];
        }
    }
    if (t != tstack) free(t);
}

/* in, out: interleaved re,im, length n; in != out */
void orc_fft_forward(const orc_fft_plan* p, const double* in, double* out) {
    fft_rec(p, 0, in, out, p->n, 1, 1);
}

/* O(n^2) long-double DFT, for known-answer tests only */
void orc_dft_naive_ld(const double* in, double* out, int n) {
    const long double twopi = 2.0L * 3.14159265358979323846264338327950288L;
    for (int k = 0; k < n; k++) {
        long double sr = 0, si = 0;
        for (int j = 0; j < n; j++) {
            long idx = ((long)j * k) % n;
            long double a = -twopi * (long double)idx / (long double)n;
            long double c = cosl(a), s = sinl(a);
            sr += (long double)in[2 * j] * c - (long double)in[2 * j + 1] * s;
            si += (long double)in[2 * j] * s + (long double)in[2 * j + 1] * c;
        }
        out[2 * k] = (double)sr;
        out[2 * k + 1] = (double)si;
    }
}

/* ------------------------------------------------------------------ */
/* dft/dft.go                                                          */
/* ------------------------------------------------------------------ */
typedef struct {
    int comp_log_pow;   /* dft.go:18 */
    double log_min;     /* dft.go:21 */
    double log_offset;  /* dft.go:24 */
    double prev_smooth; /* dft.go:27 */
    double cur_smooth;  /* dft.go:30 */
} orc_dft_params;

/* dft/dft.go:33-39 Defaults */
void orc_dft_defaults(orc_dft_params* d) {
    d->prev_smooth = 0;
    d->cur_smooth = 1.0 - d->prev_smooth;
    d->comp_log_pow = 1;
    d->log_offset = 1.0;
    d->log_min = -100;
}

/* dft/dft.go:62-85 Power.  power/log_power: [H]; *_seg: [H, T] row-major */
static void dft_power(const orc_dft_params* d, int step, int win_samples, const double* coefs,
                      double* power, double* log_power, double* power_seg, double* log_power_seg,
                      int T) {
    for (int k = 0; k < win_samples / 2 + 1; k++) {
        double rl = coefs[2 * k];
        double im = coefs[2 * k + 1];
        double powr = rl * rl + im * im;
        if (step > 0) powr = d->prev_smooth * power[k] + d->cur_smooth * powr;
        power[k] = powr;
        power_seg[(size_t)k * T + step] = powr;
        if (d->comp_log_pow) {
            double logp;
            powr += d->log_offset;
            if (powr == 0)
                logp = d->log_min;
            else
                logp = log(powr);
            log_power[k] = logp;
            log_power_seg[(size_t)k * T + step] = logp;
        }
    }
}

/* dft/dft.go:42-59 Filter + FftReal.  plan==NULL => plan rebuilt for this call,
 * as dft.go:45 does every frame ("faithful" flavour). */
int orc_dft_filter(const orc_dft_params* d, const orc_fft_plan* plan, int step,
                   const double* window, int win_samples, double* power, double* log_power,
                   double* power_seg, double* log_power_seg, int T) {
    if (win_samples < 1) return ORC_EINVAL;
    double* c = (double*)calloc(2 * (size_t)win_samples, sizeof(double));   /* dft.go:43 */
    double* o = (double*)calloc(2 * (size_t)win_samples, sizeof(double));
    if (!c || !o) {
        free(c);
        free(o);
        return ORC_EINVAL;
    }
    for (int i = 0; i < win_samples; i++) { /* dft.go:53-59 */
        c[2 * i] = window[i];
        c[2 * i + 1] = 0;
    }
    orc_fft_plan* own = NULL;
    if (!plan) {
        own = orc_fft_plan_create(win_samples); /* dft.go:45 */
        plan = own;
    }
    orc_fft_forward(plan, c, o); /* dft.go:46 */
    dft_power(d, step, win_samples, o, power, log_power, power_seg, log_power_seg, T);
    if (own) orc_fft_plan_destroy(own);
    free(c);
    free(o);
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* mel/mel.go                                                          */
/* ------------------------------------------------------------------ */
typedef struct {
    int n_filters;       /* mel.go:19 */
    double lo_hz;        /* mel.go:22 */
    double hi_hz;        /* mel.go:25 */
    double log_off;      /* mel.go:28 */
    double log_min;      /* mel.go:31 */
    int renorm;          /* mel.go:34 */
    double renorm_min;   /* mel.go:37 */
    double renorm_max;   /* mel.go:40 */
    double renorm_scale; /* mel.go:43 */
} orc_mel_fbank;

/* mel/mel.go:171-180 FilterBank.Defaults */
void orc_mel_defaults(orc_mel_fbank* m) {
    m->lo_hz = 0;
    m->hi_hz = 8000.0;
    m->n_filters = 32;
    m->log_off = 0.0;
    m->log_min = -10.0;
    m->renorm = 1;
    m->renorm_min = -6.0;
    m->renorm_max = 4.0;
    m->renorm_scale = 0.0; /* Go zero value; never set unless Renorm survives InitFilters */
}

/* mel/mel.go:156-168 */
double orc_freq_to_mel(double freq) { return 1127.0 * log(1.0 + freq / 700.0); }
double orc_mel_to_freq(double mel) { return 700.0 * (exp(mel / 1127.0) - 1.0); }
int orc_freq_to_bin(double freq, double n_fft, double sample_rate) {
    return (int)floor(((n_fft + 1) * freq) / sample_rate);
}

/* mel/mel.go:77-117 InitFilters.
 * bin_pts: [nf+2] int32; hz_pts: [nf+2]; filters: [nf, nf+2] row-major, written
 * through etensor's flat offset f*(nf+2)+fi with no per-dimension bounds check
 * (a wide triangle spills into the next row, which that row's own loop then
 * overwrites; an offset past the end of Values panics -> ORC_EPANIC). */
int orc_mel_init_filters(orc_mel_fbank* m, int dft_size, int sample_rate, int32_t* bin_pts,
                         double* hz_pts, double* filters) {
    const int nf = m->n_filters;
    const int max_bins = nf + 2;
    m->renorm = 0; /* mel.go:80 */
    double hi_mel = orc_freq_to_mel(m->hi_hz);
    double lo_mel = orc_freq_to_mel(m->lo_hz);
    double incr = (hi_mel - lo_mel) / (double)(nf + 1);
    for (int i = 0; i < max_bins; i++) {
        double ml = lo_mel + (double)i * incr;
        double hz = orc_mel_to_freq(ml);
        hz_pts[i] = hz;
        bin_pts[i] = (int32_t)orc_freq_to_bin(hz, (double)dft_size, (double)sample_rate);
    }
    const long total = (long)nf * max_bins;
    memset(filters, 0, sizeof(double) * (size_t)total); /* SetShape zero-fills */
    for (int f = 0; f < nf; f++) {
        int bin_min = bin_pts[f];
        int bin_ctr = bin_pts[f + 1];
        int bin_max = bin_pts[f + 2];
        double pkmin = (double)bin_ctr - (double)bin_min;
        double pkmax = (double)bin_max - (double)bin_ctr;
        int fi = 0;
        int bin = 0;
        for (bin = bin_min; bin <= bin_ctr; bin++, fi++) {
            double fval = ((double)bin - (double)bin_min) / pkmin; /* 0/0 = NaN when degenerate */
            long off = (long)f * max_bins + fi;
            if (off >= total) return ORC_EPANIC;
            filters[off] = fval;
        }
        for (; bin <= bin_max; bin++, fi++) {
            double fval = ((double)bin_max - (double)bin) / pkmax;
            long off = (long)f * max_bins + fi;
            if (off >= total) return ORC_EPANIC;
            filters[off] = fval;
        }
    }
    return ORC_OK;
}

/* mel/mel.go:120-153 FilterDft.  power: [H]; segment: [nf, T]; fbank: [nf] */
int orc_mel_filter_dft(const orc_mel_fbank* m, const int32_t* bin_pts, int step,
                       const double* power, int H, double* segment, int T, double* fbank,
                       const double* filters) {
    const int nf = m->n_filters;
    const long total = (long)nf * (nf + 2);
    int mi = 0;
    for (int flt = 0; flt < nf; flt++, mi++) {
        int32_t min_bin = bin_pts[flt];
        int32_t max_bin = bin_pts[flt + 2];
        double sum = 0.0;
        int fi = 0;
        for (int32_t bin = min_bin; bin <= max_bin; bin++, fi++) {
            long off = (long)mi * (nf + 2) + fi;
            if (off >= total) return ORC_EPANIC;
            if (bin < 0 || bin >= H) return ORC_EPANIC;
            double fval = filters[off];
            double pval = power[bin];
            sum += fval * pval;
        }
        sum += m->log_off;
        double val;
        if (sum == 0)
            val = m->log_min;
        else
            val = log(sum);
        if (m->renorm) {
            val -= m->renorm_min;
            if (val < 0.0) val = 0.0;
            val *= m->renorm_scale;
            if (val > 1.0) val = 1.0;
        }
        fbank[mi] = val;
        segment[(size_t)mi * T + step] = val;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* agabor/gabor.go                                                     */
/* ------------------------------------------------------------------ */
typedef struct {
    int off;             /* gabor.go:20 */
    double wave_len;     /* gabor.go:23 */
    double orientation;  /* gabor.go:26 */
    double sigma_width;  /* gabor.go:29 */
    double sigma_length; /* gabor.go:32 */
    double phase_offset; /* gabor.go:35 */
    int circle_edge;     /* gabor.go:38 */
    int circular;        /* gabor.go:41 */
} orc_gabor_spec;

/* agabor/gabor.go:329-336 Active: returns count, writes compacted specs */
int orc_gabor_active(const orc_gabor_spec* specs, int n, orc_gabor_spec* active) {
    int c = 0;
    for (int i = 0; i < n; i++)
        if (!specs[i].off) active[c++] = specs[i];
    return c;
}

/* agabor/gabor.go:89-222 ToTensor (incl. Filter.Defaults :73-86).
 * specs: the ACTIVE specs; out: [n, sy, sx] float64 */
void orc_gabor_to_tensor(const orc_gabor_spec* specs, int n, int sx, int sy, int distribute,
                         double* out) {
    int nhf = 0, nvf = 0;
    if (distribute) {
        for (int i = 0; i < n; i++) {
            if (specs[i].orientation == 0)
                nhf++;
            else if (specs[i].orientation == 90)
                nvf++;
        }
    } else {
        nhf = 1;
        nvf = 1;
    }
    double radius_x = (double)sx / 2.0;
    double radius_y = (double)sy / 2.0;
    double ctr_x = (double)(sx - 1) / 2.0;
    double ctr_y = (double)(sy - 1) / 2.0;
    double h_ctr_inc = (double)(sy - 1) / (double)(nhf + 1);
    double v_ctr_inc = (double)(sx - 1) / (double)(nvf + 1);
    int h_cnt = 0, v_cnt = 0;
    for (int i = 0; i < n; i++) {
        orc_gabor_spec f = specs[i];
        /* gabor.go:73-86 Defaults */
        if (f.wave_len == 0) f.wave_len = 2;
        if (f.sigma_length == 0 && !f.circular) f.sigma_length = 0.5;
        if (f.sigma_width == 0) f.sigma_width = 0.5;
        double two_pi_norm = (2.0 * M_PI) / f.wave_len;
        double l_norm = 1.0 / (2.0 * f.sigma_length * f.sigma_length);
        double w_norm = 1.0 / (2.0 * f.sigma_width * f.sigma_width);
        double h_pos = 0, v_pos = 0;
        if (distribute) {
            if (f.orientation == 0) {
                h_pos = h_ctr_inc * (double)(h_cnt + 1);
                h_cnt++;
            }
            if (f.orientation == 90) {
                v_pos = v_ctr_inc * (double)(v_cnt + 1);
                v_cnt++;
            }
        } else {
            h_pos = h_ctr_inc * (double)(h_cnt + 1);
            v_pos = v_ctr_inc * (double)(v_cnt + 1);
        }
        double* o = out + (size_t)i * sy * sx;
        if (!f.circular) {
            for (int y = 0; y < sy; y++) {
                for (int x = 0; x < sx; x++) {
                    double xf = (double)x - ctr_x;
                    double yf = (double)y - ctr_y;
                    if (f.orientation == 0) yf = (double)y - h_pos;
                    if (f.orientation == 90) xf = (double)x - v_pos;
                    double xfn = xf / radius_x;
                    double yfn = yf / radius_y;
                    double dist = hypot(xfn, yfn);
                    double val = 0;
                    if (!(f.circle_edge && dist > 1.0)) {
                        double radians = f.orientation * M_PI / 180;
                        double nx = xfn * cos(radians) - yfn * sin(radians);
                        double ny = yfn * cos(radians) + xfn * sin(radians);
                        double gauss = exp(-(w_norm * (nx * nx) + l_norm * (ny * ny)));
                        double sin_val = sin(two_pi_norm * ny + f.phase_offset);
                        val = gauss * sin_val;
                    }
                    o[y * sx + x] = val;
                }
            }
        } else { /* circular, gabor.go:172-191 */
            double norm = 1.0 / (2.0 * f.sigma_width * f.sigma_width);
            for (int y = 0; y < sy; y++) {
                for (int x = 0; x < sx; x++) {
                    double xf = (double)x - ctr_x;
                    double yf = (double)y - ctr_y;
                    double xfn = xf / radius_x;
                    double yfn = yf / radius_y;
                    double nx = xfn * xfn * norm;
                    double ny = yfn * yfn * norm;
                    double gauss = sqrt(nx + ny);
                    double sin_val = sin(two_pi_norm * nx * ny);
                    o[y * sx + x] = -gauss * sin_val;
                }
            }
        }
    }
    /* renorm each half, gabor.go:195-221 */
    for (int i = 0; i < n; i++) {
        double* o = out + (size_t)i * sy * sx;
        double pos_sum = 0.0, neg_sum = 0.0;
        for (int j = 0; j < sy * sx; j++) {
            double val = o[j];
            if (val > 0)
                pos_sum += val;
            else if (val < 0)
                neg_sum += val;
        }
        double pos_norm = 1.0 / pos_sum;
        double neg_norm = -1.0 / neg_sum;
        for (int j = 0; j < sy * sx; j++) {
            double val = o[j];
            if (val > 0.0)
                val *= pos_norm;
            else if (val < 0.0)
                val *= neg_norm;
            o[j] = val;
        }
    }
}

/* agabor/gabor.go:225-315 Convolve.
 * mel: [nf, T] float64 (etensor flat offset, no per-dim bounds check);
 * k: [n_g, sy, sx]; out: float32, rank 2 ([d0, d1]) or 4 ([d0, d1, d2, d3]),
 * in/out -- cells the reference does not write keep their contents. */
int orc_gabor_convolve(const double* mel, int nf, int T, const double* k, int n_g, int sx, int sy,
                       int stride_x, int stride_y, double gain, float* out, int out_rank,
                       const int* out_shape, int by_time) {
    if (T < sx) return ORC_EINVAL; /* gabor.go:226-229 */
    int t_max = 1, f_max = 1, t_max_strides = 1;
    long out_total = 1;
    for (int i = 0; i < out_rank; i++) out_total *= out_shape[i];
    if (out_rank == 2) { /* gabor.go:234-250 */
        int x = T - sx;
        if (x == 0 || x < stride_x) {
        } else {
            t_max = x + 1;
        }
        int z = T - sx;
        t_max_strides = z / stride_x + 1;
        int y = nf - sy;
        if (y == 0 || y < stride_y) {
        } else {
            f_max = y + 1;
        }
    } else if (out_rank == 4) { /* gabor.go:251-258 */
        int t_max1 = out_shape[1] * stride_x;
        int t_max2 = T - stride_x;
        t_max = (int)fmin((double)t_max1, (double)t_max2);
        int f_max1 = out_shape[0] * stride_y;
        int f_max2 = nf - stride_y;
        f_max = (int)fmin((double)f_max1, (double)f_max2);
    } else {
        return ORC_EINVAL; /* gabor.go:259-262 */
    }
    const long mel_total = (long)nf * T;
    int t_idx = 0;
    for (int t = 0; t < t_max; t += stride_x, t_idx++) {
        int f_idx = 0;
        for (int f = 0; f < f_max; f += stride_y, f_idx++) {
            for (int flt = 0; flt < n_g; flt++) {
                double f_sum = 0.0;
                for (int ff = 0; ff < sy; ff++) {
                    for (int ft = 0; ft < sx; ft++) {
                        double f_val = k[((size_t)flt * sy + ff) * sx + ft];
                        long off = (long)(f + ff) * T + (t + ft);
                        if (off >= mel_total) return ORC_EPANIC;
                        double i_val = mel[off];
                        if (isnan(i_val)) i_val = .5;
                        f_sum += f_val * i_val;
                    }
                }
                int pos = f_sum >= 0.0;
                double act = gain * fabs(f_sum);
                if (out_rank == 2) {
                    int y = f_idx * 2;
                    int x;
                    if (by_time)
                        x = t_idx + t_max_strides * flt;
                    else
                        x = flt + t_idx * n_g;
                    long o0 = (long)y * out_shape[1] + x;
                    long o1 = (long)(y + 1) * out_shape[1] + x;
                    if (o0 >= out_total || o1 >= out_total) return ORC_EPANIC;
                    if (pos) {
                        out[o0] = (float)act;
                        out[o1] = 0;
                    } else {
                        out[o0] = 0;
                        out[o1] = (float)act;
                    }
                } else {
                    long s2 = out_shape[3], s1 = (long)out_shape[2] * s2, s0 = out_shape[1] * s1;
                    long o0 = f_idx * s0 + t_idx * s1 + 0 * s2 + flt;
                    long o1 = f_idx * s0 + t_idx * s1 + 1 * s2 + flt;
                    if (o0 >= out_total || o1 >= out_total) return ORC_EPANIC;
                    if (pos) {
                        out[o0] = (float)act;
                        out[o1] = 0;
                    } else {
                        out[o0] = 0;
                        out[o1] = (float)act;
                    }
                }
            }
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* sound/sndenv.go segment loop                                        */
/* ------------------------------------------------------------------ */

/* sound/sndenv.go:455-478 SndToWindow: fills window[N]; ORC_ESHORT if end > len */
int orc_snd_to_window(const double* signal, long sig_len, long start, int win_samples,
                      double* window) {
    long end = start + win_samples;
    if (end > sig_len) return ORC_ESHORT;
    if (start < 0 && end <= 0) {
        for (int i = 0; i < win_samples; i++) window[i] = 0;
    } else if (start < 0 && end > 0) {
        long npad = -start;
        for (long i = 0; i < npad; i++) window[i] = 0;
        for (long i = 0; i < end; i++) window[npad + i] = signal[i];
    } else {
        for (int i = 0; i < win_samples; i++) window[i] = signal[start + i];
    }
    return ORC_OK;
}

typedef struct {
    int sample_rate;
    int win_samples;    /* N  */
    int step_samples;   /* S  */
    int stride_samples; /*    */
    int segment_steps;  /* T  */
    int border_steps;
} orc_snd_params;

/* sound/sndenv.go:342-359 ProcessSegment (loop part) + :438-452 ProcessStep.
 * Outputs (all zeroed first, :343-351): power[H], log_power[H], power_seg[H,T],
 * log_power_seg[H,T], mel_seg[nf,T].  fbank[nf] is the per-step MelFBank.
 * Stops at the first ESHORT frame, leaving later columns zero (:354-358).
 * faithful != 0 => FFT plan rebuilt every frame (dft.go:45).
 * Returns the number of frames processed. */
int orc_process_segment(const orc_snd_params* sp, const orc_dft_params* dp,
                        const orc_mel_fbank* mp, const int32_t* bin_pts, const double* filters,
                        const double* signal, long sig_len, int segment, int add_ms,
                        int faithful, double* power, double* log_power, double* power_seg,
                        double* log_power_seg, double* mel_seg, double* fbank) {
    const int N = sp->win_samples, T = sp->segment_steps, H = N / 2 + 1, nf = mp->n_filters;
    memset(power, 0, sizeof(double) * H);
    memset(log_power, 0, sizeof(double) * H);
    memset(power_seg, 0, sizeof(double) * (size_t)H * T);
    memset(log_power_seg, 0, sizeof(double) * (size_t)H * T);
    memset(mel_seg, 0, sizeof(double) * (size_t)nf * T);
    double* window = (double*)malloc(sizeof(double) * N);
    orc_fft_plan* plan = faithful ? NULL : orc_fft_plan_create(N);
    int done = 0;
    for (int s = 0; s < T; s++) {
        /* :247-251 Steps[i] = StepSamples*(i - BorderSteps); :440-441 */
        long offset = (long)sp->step_samples * (s - sp->border_steps) +
                      orc_msec_to_samples((double)add_ms, sp->sample_rate);
        long start = (long)segment * sp->stride_samples + offset;
        int err = orc_snd_to_window(signal, sig_len, start, N, window);
        if (err != ORC_OK) break;
        orc_dft_filter(dp, plan, s, window, N, power, log_power, power_seg, log_power_seg, T);
        err = orc_mel_filter_dft(mp, bin_pts, s, power, H, mel_seg, T, fbank, filters);
        if (err != ORC_OK) {
            done = -err;
            break;
        }
        done++;
    }
    if (plan) orc_fft_plan_destroy(plan);
    free(window);
    return done;
}

/* ------------------------------------------------------------------ */
/* MFCC tail of the segment loop                         */
/* ------------------------------------------------------------------ */

/* gonum v0.11.0 dsp/fourier DCT.Transform (FFTPACK `cost`, not vendored in the reference; contract
 * restated): unnormalised DCT-I,
 *   y[k] = x[0] + (-1)^k x[n-1] + 2 sum_{j=1}^{n-2} x[j] cos(pi j k / (n-1)),  n > 1.        */
void orc_dct1(const double* x, double* y, int n) {
    for (int k = 0; k < n; k++) {
        long double s = (long double)x[0] + ((k & 1) ? -(long double)x[n - 1] : (long double)x[n - 1]);
        for (int j = 1; j < n - 1; j++)
            s += 2.0L * (long double)x[j] * cosl(3.14159265358979323846264338327950288L * (long double)j * (long double)k /
                                                 (long double)(n - 1));
        y[k] = (double)s;
    }
}

/* mel/mel.go:192-212 CepstrumDct: DCT of the nf log-mel values of one step, c0 <- ln(1 + c0^2),
 * first n_coefs into mfcc_seg[n_coefs, T] column `step`. */
void orc_cepstrum_dct(int step, const double* fbank, int nf, double* mfcc_seg, int n_coefs, int T) {
    double* out = (double*)malloc(sizeof(double) * (size_t)nf);
    orc_dct1(fbank, out, nf);          /* mel.go:198-202 */
    double el0 = out[0];
    out[0] = log(1.0 + el0 * el0);     /* mel.go:203-204 */
    for (int i = 0; i < n_coefs; i++) mfcc_seg[(size_t)i * T + step] = out[i]; /* :207-209 */
    free(out);
}

/* sound/sndenv.go:360-432: Energy (with the reference's axis quirk, SURVEY Q8: Energy[s] sums
 * LogPowerSegment.FloatValRowCell(s, f) = Values[s*T + f] over f < T, i.e. the log-power of BIN s
 * over the steps), the row-0 overwrite, and the delta / delta-delta passes whose running sums are
 * not reset per coefficient.  Returns ORC_EPANIC where the Go code would index out of range. */
int orc_mfcc_tail(const double* log_power_seg, int H, int T, double* energy, double* mfcc_seg,
                  int n_coefs, int deltas, double* mfcc_deltas, double* mfcc_delta_deltas) {
    for (int s = 0; s < T; s++) {                     /* :360-366 */
        double e = 0.0;
        for (int f = 0; f < T; f++) {
            long idx = (long)s * T + f;
            if (idx >= (long)H * T) return ORC_EPANIC;
            e += log_power_seg[idx];
        }
        energy[s] = e;
    }
    for (int s = 0; s < T; s++) mfcc_seg[s] = energy[s]; /* :368-372 SetFloatRowCell(0, s, e) */
    if (!deltas) return ORC_OK;
    const int npn = 2;
    for (int pass = 0; pass < 2; pass++) {            /* :378-404 deltas, :407-431 delta-deltas */
        const double* src = pass == 0 ? mfcc_seg : mfcc_deltas;
        double* dst = pass == 0 ? mfcc_deltas : mfcc_delta_deltas;
        for (int s = 0; s < T; s++) {
            double prv = 0.0, nxt = 0.0;
            for (int i = 0; i < n_coefs; i++) {
                double nume = 0.0;
                for (int n = 1; n <= npn; n++) {
                    int sprv = s - n, snxt = s + n;
                    if (sprv < 0) sprv = 0;
                    if (snxt > T - 1) snxt = T - 1;
                    prv += src[(size_t)i * T + sprv];
                    nxt += src[(size_t)i * T + snxt];
                    nume += (double)n * (nxt - prv);
                    double denom = (double)(2 * n * n);
                    dst[(size_t)i * T + s] = nume / denom;
                }
            }
        }
    }
    return ORC_OK;
}

/* ProcessSegment with the MFCC tail on (mel.Params.MFCC = true, the SndEnv default):
 * the loop of orc_process_segment plus CepstrumDct per processed step (sndenv.go:447-449)
 * and orc_mfcc_tail.  mfcc_seg / deltas / delta_deltas: [n_coefs, T]; energy: [T]. */
int orc_process_segment_mfcc(const orc_snd_params* sp, const orc_dft_params* dp,
                             const orc_mel_fbank* mp, const int32_t* bin_pts, const double* filters,
                             const double* signal, long sig_len, int segment, int add_ms,
                             double* power, double* log_power, double* power_seg,
                             double* log_power_seg, double* mel_seg, double* fbank, int n_coefs,
                             int deltas, double* energy, double* mfcc_seg, double* mfcc_deltas,
                             double* mfcc_delta_deltas) {
    const int N = sp->win_samples, T = sp->segment_steps, H = N / 2 + 1, nf = mp->n_filters;
    memset(power, 0, sizeof(double) * H);
    memset(log_power, 0, sizeof(double) * H);
    memset(power_seg, 0, sizeof(double) * (size_t)H * T);
    memset(log_power_seg, 0, sizeof(double) * (size_t)H * T);
    memset(mel_seg, 0, sizeof(double) * (size_t)nf * T);
    memset(energy, 0, sizeof(double) * T);
    memset(mfcc_seg, 0, sizeof(double) * (size_t)n_coefs * T); /* :349-351 */
    /* MFCCDeltas / MFCCDeltaDeltas are NOT zeroed per segment by the reference; every cell is rewritten */
    double* window = (double*)malloc(sizeof(double) * N);
    orc_fft_plan* plan = orc_fft_plan_create(N);
    int done = 0;
    for (int s = 0; s < T; s++) {
        long offset = (long)sp->step_samples * (s - sp->border_steps) +
                      orc_msec_to_samples((double)add_ms, sp->sample_rate);
        long start = (long)segment * sp->stride_samples + offset;
        if (orc_snd_to_window(signal, sig_len, start, N, window) != ORC_OK) break;
        orc_dft_filter(dp, plan, s, window, N, power, log_power, power_seg, log_power_seg, T);
        int err = orc_mel_filter_dft(mp, bin_pts, s, power, H, mel_seg, T, fbank, filters);
        if (err != ORC_OK) {
            done = -err;
            break;
        }
        orc_cepstrum_dct(s, fbank, nf, mfcc_seg, n_coefs, T);
        done++;
    }
    orc_fft_plan_destroy(plan);
    free(window);
    if (done < 0) return done;
    int rc = orc_mfcc_tail(log_power_seg, H, T, energy, mfcc_seg, n_coefs, deltas, mfcc_deltas,
                           mfcc_delta_deltas);
    return rc == ORC_OK ? done : -rc;
}

/* Batch driver for the CPU baseline (bench.py cpu_baseline leg) and for
 * batch-level parity tests: n_items independent (signal, segment) pairs, each
 * through orc_process_segment and optionally orc_gabor_convolve (4-D).
 * sig: flat float64; sig_off/sig_len per item; seg per item.
 * mel_out: [n_items, nf, T] float64; gabor_out: [n_items, PY, PX, 2, nG] float32 or NULL. */
int orc_process_batch(const orc_snd_params* sp, const orc_dft_params* dp, const orc_mel_fbank* mp,
                      const int32_t* bin_pts, const double* filters, const double* sig,
                      const int64_t* sig_off, const int32_t* sig_len, const int32_t* seg,
                      int n_items, int faithful, double* mel_out, const double* gabor_k, int n_g,
                      int sx, int sy, int stride_x, int stride_y, double gain, int py, int px,
                      float* gabor_out) {
    const int N = sp->win_samples, T = sp->segment_steps, H = N / 2 + 1, nf = mp->n_filters;
    double* power = (double*)malloc(sizeof(double) * H);
    double* log_power = (double*)malloc(sizeof(double) * H);
    double* power_seg = (double*)malloc(sizeof(double) * (size_t)H * T);
    double* log_power_seg = (double*)malloc(sizeof(double) * (size_t)H * T);
    double* fbank = (double*)malloc(sizeof(double) * nf);
    int rc = ORC_OK;
    for (int i = 0; i < n_items; i++) {
        double* mel = mel_out + (size_t)i * nf * T;
        int done = orc_process_segment(sp, dp, mp, bin_pts, filters, sig + sig_off[i], sig_len[i],
                                       seg[i], 0, faithful, power, log_power, power_seg,
                                       log_power_seg, mel, fbank);
        if (done < 0) {
            rc = -done;
            break;
        }
        if (gabor_out) {
            int shape[4] = {py, px, 2, n_g};
            float* go = gabor_out + (size_t)i * py * px * 2 * n_g;
            rc = orc_gabor_convolve(mel, nf, T, gabor_k, n_g, sx, sy, stride_x, stride_y, gain, go,
                                    4, shape, 0);
            if (rc != ORC_OK) break;
        }
    }
    free(power);
    free(log_power);
    free(power_seg);
    free(log_power_seg);
    free(fbank);
    return rc;
}
