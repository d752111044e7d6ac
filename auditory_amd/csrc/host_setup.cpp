// Host-side setup of the hot path: parameter derivation, mel filter table, gabor kernels.
// These run once per plan on the CPU (in the reference they run once per SndEnv.Init) and
// their float64 results are uploaded to the device by aud_plan_create.  No GPU calls here.
#include <cmath>
#include <cstring>
#include <vector>

#include "auditory_hip.h"

namespace {
constexpr double kPi = 3.14159265358979323846;

// Go's math.Round rounds half away from zero; std::round does the same.
inline int go_round_to_int(double v) { return static_cast<int>(std::round(v)); }
}  // namespace

extern "C" {

int aud_version(void) { return AUD_VERSION; }

const char* aud_status_string(int status) {
    switch (status) {
        case AUD_OK: return "ok";
        case AUD_EINVAL: return "invalid argument or shape outside the supported envelope";
        case AUD_EHIP: return "HIP runtime error";
        case AUD_ERCCL: return "RCCL error";
        case AUD_ENOMEM: return "out of memory";
        case AUD_ESHORT: return "SndToWindow: end beyond signal length!!";
        case AUD_EBROKEN: return "the direct gather is broken (a peer never arrived, or a call failed half-way): destroy and create it again";
        default: return "unknown status";
    }
}

// sound/sndenv.go:522-524
int aud_msec_to_samples(double ms, int rate) { return go_round_to_int(ms * 0.001 * double(rate)); }

// sound/sndenv.go:64-71
void aud_sound_params_defaults(aud_sound_params* p) {
    std::memset(p, 0, sizeof(*p));
    p->win_ms = 25.0;
    p->step_ms = 10.0;
    p->segment_ms = 100.0;
    p->channel = 0;
    p->stride_ms = 100.0;
    p->border_steps = 2;
}

// sound/sndenv.go:195-207
int aud_sound_params_derive(aud_sound_params* p, int sample_rate) {
    if (sample_rate <= 0) return AUD_EINVAL;  // "sample rate <= 0"
    p->win_samples = aud_msec_to_samples(p->win_ms, sample_rate);
    p->step_samples = aud_msec_to_samples(p->step_ms, sample_rate);
    p->segment_samples = aud_msec_to_samples(p->segment_ms, sample_rate);
    p->segment_steps = go_round_to_int(p->segment_ms / p->step_ms) + 2 * p->border_steps;
    p->stride_samples = aud_msec_to_samples(p->stride_ms, sample_rate);
    return AUD_OK;
}

// sound/sndenv.go:263-265
int aud_seg_cnt(int signal_len, int segment_samples, int stride_samples, int channels) {
    int rest = (signal_len - segment_samples * channels) / channels;
    return rest / stride_samples + 1;
}

// sound/sndenv.go:503-507
int aud_tail(int signal_len, int segment_samples, int stride_samples) {
    return (signal_len - segment_samples) % stride_samples;
}

// sound/sndenv.go:510-519 (length of the pad that Pad appends)
int aud_pad_len(int signal_len, int segment_samples, int stride_samples, int step_samples) {
    int tail = aud_tail(signal_len, segment_samples, stride_samples);
    return segment_samples - step_samples - tail % step_samples;
}

// sound/sndenv.go:274-294
int aud_adjust_for_silence(double add_ms, double existing_ms, int sample_rate, int* delta_samples) {
    if (delta_samples) *delta_samples = 0;
    if (sample_rate <= 0) return -1;  // "sample rate <= 0"
    int offset = 0;
    if (add_ms >= 0) {
        if (add_ms < existing_ms) {
            offset = static_cast<int>(existing_ms - add_ms);  // Go int(): truncation
            if (delta_samples) *delta_samples = -aud_msec_to_samples(double(offset), sample_rate);
        } else if (add_ms > existing_ms) {
            offset = static_cast<int>(add_ms - existing_ms);
            if (delta_samples) *delta_samples = aud_msec_to_samples(double(offset), sample_rate);
        }
    }
    return offset;
}

// sound/sound.go:130-141
double aud_pcm_to_float(int value, int bit_depth) {
    switch (bit_depth) {
        case 32: return double(value) / double(0x7FFFFFFF);
        case 24: return double(value) / double(0x7FFFFF);
        case 16: return double(value) / double(0x7FFF);
        case 8: return double(value) / double(0x7F);
        default: return 0.0;
    }
}

// dft/dft.go:33-39
void aud_dft_defaults(aud_dft_params* d) {
    d->prev_smooth = 0.0;
    d->cur_smooth = 1.0 - d->prev_smooth;
    d->comp_log_pow = 1;
    d->log_offset = 1.0;
    d->log_min = -100.0;
}

// mel/mel.go:171-180
void aud_mel_defaults(aud_mel_fbank* m) {
    m->lo_hz = 0.0;
    m->hi_hz = 8000.0;
    m->n_filters = 32;
    m->log_off = 0.0;
    m->log_min = -10.0;
    m->renorm = 1;
    m->renorm_min = -6.0;
    m->renorm_max = 4.0;
    m->renorm_scale = 0.0;
}

// mel/mel.go:156-168
double aud_freq_to_mel(double freq) { return 1127.0 * std::log(1.0 + freq / 700.0); }
double aud_mel_to_freq(double mel) { return 700.0 * (std::exp(mel / 1127.0) - 1.0); }
int aud_freq_to_bin(double freq, double n_fft, double sample_rate) {
    return static_cast<int>(std::floor(((n_fft + 1.0) * freq) / sample_rate));
}

// mel/mel.go:77-117.  The table is addressed the way etensor addresses it: one flat
// row-major offset row*(nf+2)+col with no per-dimension check, so an over-wide triangle
// runs on into the next row (and is then overwritten by that row's own triangle); only an
// offset past the end of the storage is an error (the Go code panics there).
int aud_mel_init_filters(aud_mel_fbank* m, int dft_size, int sample_rate, int32_t* bin_pts,
                         double* hz_pts, double* filters) {
    if (!m || !bin_pts || !filters || m->n_filters <= 0 || dft_size <= 0 || sample_rate <= 0)
        return AUD_EINVAL;
    const int nf = m->n_filters;
    const int cols = nf + 2;
    m->renorm = 0;  // mel.go:80

    const double mel_hi = aud_freq_to_mel(m->hi_hz);
    const double mel_lo = aud_freq_to_mel(m->lo_hz);
    const double mel_step = (mel_hi - mel_lo) / double(nf + 1);
    for (int i = 0; i < cols; ++i) {
        const double hz = aud_mel_to_freq(mel_lo + double(i) * mel_step);
        if (hz_pts) hz_pts[i] = hz;
        bin_pts[i] = int32_t(aud_freq_to_bin(hz, double(dft_size), double(sample_rate)));
    }

    const size_t cells = size_t(nf) * size_t(cols);
    std::fill(filters, filters + cells, 0.0);
    for (int f = 0; f < nf; ++f) {
        const int lo = bin_pts[f], ctr = bin_pts[f + 1], hi = bin_pts[f + 2];
        const double rise = double(ctr) - double(lo);
        const double fall = double(hi) - double(ctr);
        size_t cell = size_t(f) * size_t(cols);
        int bin = lo;
        for (; bin <= ctr; ++bin, ++cell) {  // rising edge, 0/0 = NaN when lo == ctr
            if (cell >= cells) return AUD_EINVAL;
            filters[cell] = (double(bin) - double(lo)) / rise;
        }
        for (; bin <= hi; ++bin, ++cell) {  // falling edge
            if (cell >= cells) return AUD_EINVAL;
            filters[cell] = (double(hi) - double(bin)) / fall;
        }
    }
    return AUD_OK;
}

// agabor/gabor.go:329-336
int aud_gabor_active(const aud_gabor_spec* specs, int n, aud_gabor_spec* active) {
    int kept = 0;
    for (int i = 0; i < n; ++i)
        if (!specs[i].off) active[kept++] = specs[i];
    return kept;
}

// agabor/gabor.go:89-222
int aud_gabor_to_tensor(const aud_gabor_spec* specs, int n, const aud_gabor_set* set, double* out,
                        int* n_out) {
    if (!specs || !set || !out || n < 0 || set->size_x <= 0 || set->size_y <= 0) return AUD_EINVAL;
    std::vector<aud_gabor_spec> act(size_t(n) + 1);
    const int na = aud_gabor_active(specs, n, act.data());
    if (n_out) *n_out = na;
    const int sx = set->size_x, sy = set->size_y;
    const size_t area = size_t(sx) * size_t(sy);

    int n_horiz = 1, n_vert = 1;
    if (set->distribute) {
        n_horiz = n_vert = 0;
        for (int i = 0; i < na; ++i) {
            if (act[i].orientation == 0) ++n_horiz;
            else if (act[i].orientation == 90) ++n_vert;
        }
    }
    const double rad_x = double(sx) / 2.0, rad_y = double(sy) / 2.0;
    const double mid_x = double(sx - 1) / 2.0, mid_y = double(sy - 1) / 2.0;
    const double h_inc = double(sy - 1) / double(n_horiz + 1);
    const double v_inc = double(sx - 1) / double(n_vert + 1);
    int h_seen = 0, v_seen = 0;

    for (int i = 0; i < na; ++i) {
        aud_gabor_spec g = act[i];
        // Filter.Defaults, gabor.go:73-86
        if (g.wave_len == 0) g.wave_len = 2;
        if (g.sigma_length == 0 && !g.circular) g.sigma_length = 0.5;
        if (g.sigma_width == 0) g.sigma_width = 0.5;

        const double k_wave = (2.0 * kPi) / g.wave_len;
        const double k_len = 1.0 / (2.0 * g.sigma_length * g.sigma_length);
        const double k_wid = 1.0 / (2.0 * g.sigma_width * g.sigma_width);

        double h_pos = 0.0, v_pos = 0.0;
        if (set->distribute) {
            if (g.orientation == 0) h_pos = h_inc * double(++h_seen);
            if (g.orientation == 90) v_pos = v_inc * double(++v_seen);
        } else {
            h_pos = h_inc * double(h_seen + 1);
            v_pos = v_inc * double(v_seen + 1);
        }

        double* dst = out + size_t(i) * area;
        for (int y = 0; y < sy; ++y) {
            for (int x = 0; x < sx; ++x) {
                double dx = double(x) - mid_x;
                double dy = double(y) - mid_y;
                double v;
                if (!g.circular) {
                    if (g.orientation == 0) dy = double(y) - h_pos;
                    if (g.orientation == 90) dx = double(x) - v_pos;
                    const double u = dx / rad_x, w = dy / rad_y;
                    v = 0.0;
                    if (!(g.circle_edge && std::hypot(u, w) > 1.0)) {
                        const double th = g.orientation * kPi / 180;
                        const double across = u * std::cos(th) - w * std::sin(th);
                        const double along = w * std::cos(th) + u * std::sin(th);
                        const double env = std::exp(-(k_wid * (across * across) + k_len * (along * along)));
                        v = env * std::sin(k_wave * along + g.phase_offset);
                    }
                } else {  // gabor.go:172-191
                    const double u = dx / rad_x, w = dy / rad_y;
                    const double a = u * u * k_wid, b = w * w * k_wid;
                    v = -std::sqrt(a + b) * std::sin(k_wave * a * b);
                }
                dst[size_t(y) * sx + x] = v;
            }
        }
    }

    // gabor.go:195-221: positive lobe scaled to sum +1, negative lobe to sum -1
    for (int i = 0; i < na; ++i) {
        double* dst = out + size_t(i) * area;
        double pos = 0.0, neg = 0.0;
        for (size_t c = 0; c < area; ++c) {
            if (dst[c] > 0) pos += dst[c];
            else if (dst[c] < 0) neg += dst[c];
        }
        const double pos_scale = 1.0 / pos, neg_scale = -1.0 / neg;
        for (size_t c = 0; c < area; ++c) {
            if (dst[c] > 0.0) dst[c] *= pos_scale;
            else if (dst[c] < 0.0) dst[c] *= neg_scale;
        }
    }
    return AUD_OK;
}

// agabor/gabor.go:226-262
int aud_gabor_iter_space(const aud_gabor_set* set, int mel_rows, int mel_cols, int out_rank,
                         const int32_t* out_shape, int32_t* n_t, int32_t* n_f,
                         int32_t* t_max_strides) {
    if (!set || set->stride_x <= 0 || set->stride_y <= 0) return AUD_EINVAL;
    if (mel_cols < set->size_x) return AUD_EINVAL;  // "Gabor filter width can not be larger ..."
    int t_max = 1, f_max = 1, strides = 1;
    if (out_rank == 2) {
        const int x = mel_cols - set->size_x;
        if (!(x == 0 || x < set->stride_x)) t_max = x + 1;
        strides = x / set->stride_x + 1;
        const int y = mel_rows - set->size_y;
        if (!(y == 0 || y < set->stride_y)) f_max = y + 1;
    } else if (out_rank == 4) {
        if (!out_shape) return AUD_EINVAL;
        t_max = int(std::fmin(double(out_shape[1] * set->stride_x), double(mel_cols - set->stride_x)));
        f_max = int(std::fmin(double(out_shape[0] * set->stride_y), double(mel_rows - set->stride_y)));
    } else {
        return AUD_EINVAL;  // "The output tensor should have 2 or 4 dimensions"
    }
    // for t := 0; t < tMax; t += StrideX  =>  ceil(tMax / StrideX) iterations (0 if tMax <= 0)
    auto trips = [](int lim, int step) { return lim <= 0 ? 0 : (lim + step - 1) / step; };
    if (n_t) *n_t = trips(t_max, set->stride_x);
    if (n_f) *n_f = trips(f_max, set->stride_y);
    if (t_max_strides) *t_max_strides = strides;
    return AUD_OK;
}

// sound/sndenv.go:527-529
double aud_samples_to_msec(int samples, int rate) { return 1000.0 * double(samples) / double(rate); }

// kwta.KWTA.Defaults() of github.com/emer/vision v1.1.15 over leabra v1.1.48's fffb.Params.Defaults()
// and nxx1.Params.Defaults().  Those sources are not in the reference tree; the values are the published
// ones as far as known here.  A Go binding never needs this function: it passes the fields of the real
// kwta.KWTA value (se.Kwta, sound/sndenv.go:175, set by se.Kwta.Defaults() at :189).
void aud_kwta_defaults(aud_kwta_params* k) {
    if (!k) return;
    std::memset(k, 0, sizeof(*k));
    k->on = 1;
    k->iters = 20;
    k->del_act_thr = 0.005f;
    const aud_fffb_params f = {1, 1.8f, 1.0f, 1.0f, 1.4f, 0.0f, 0.1f};
    k->lay_fffb = f;
    k->pool_fffb = f;
    k->pool_fffb.gi = 2.0f;
    const aud_nxx1_params x = {0.5f, 80.0f, 0.01f, 0.01f, 0.33f, 0.8f, 3.0f, 0.01f, 10.0f, 0.1f};
    k->xx1 = x;
    k->act_tau = 3.0f;
    const float gbar[4] = {0.5f, 0.1f, 1.0f, 1.0f}, erev[4] = {1.0f, 0.3f, 0.25f, 0.1f};
    std::memcpy(k->gbar, gbar, sizeof gbar);
    std::memcpy(k->erev, erev, sizeof erev);
}

}  // extern "C"
