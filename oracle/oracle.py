"""ctypes binding of oracle/liboracle.so (the CPU float64 restatement).

TEST INFRASTRUCTURE ONLY -- see the header of auditory_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under auditory_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORC_OK, ORC_EINVAL, ORC_EPANIC, ORC_ESHORT = 0, 1, 2, 3


def build(force=False):
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:  # parallel test workers build once, not at once
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(force)


def _build(force):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("auditory_oracle.c", "kwta_oracle.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


class DftParams(C.Structure):
    _fields_ = [("comp_log_pow", C.c_int), ("log_min", C.c_double), ("log_offset", C.c_double),
                ("prev_smooth", C.c_double), ("cur_smooth", C.c_double)]


class MelFBank(C.Structure):
    _fields_ = [("n_filters", C.c_int), ("lo_hz", C.c_double), ("hi_hz", C.c_double),
                ("log_off", C.c_double), ("log_min", C.c_double), ("renorm", C.c_int),
                ("renorm_min", C.c_double), ("renorm_max", C.c_double),
                ("renorm_scale", C.c_double)]


class GaborSpec(C.Structure):
    _fields_ = [("off", C.c_int), ("wave_len", C.c_double), ("orientation", C.c_double),
                ("sigma_width", C.c_double), ("sigma_length", C.c_double),
                ("phase_offset", C.c_double), ("circle_edge", C.c_int), ("circular", C.c_int)]


class Fffb(C.Structure):
    _fields_ = [("on", C.c_int32)] + [(n, C.c_float) for n in ("gi", "ff", "fb", "fb_tau", "max_vs_avg", "ff0")]


class Nxx1(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("thr", "gain", "nvar", "vm_act_thr", "sig_mult", "sig_mult_pow",
                                         "sig_gain", "interp_range", "gain_cor_range", "gain_cor")]


class Kwta(C.Structure):
    _fields_ = [("on", C.c_int32), ("iters", C.c_int32), ("del_act_thr", C.c_float), ("lay", Fffb),
                ("pool", Fffb), ("xx1", Nxx1), ("act_tau", C.c_float), ("gbar", C.c_float * 4),
                ("erev", C.c_float * 4)]


class KwtaDerived(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lay_fb_dt", "pool_fb_dt", "sig_gain_nvar", "sig_mult_eff",
                                         "sig_val_at0", "interp_val")] + \
               [("erev_sub_thr", C.c_float * 4), ("thr_sub_erev", C.c_float * 4), ("act_dt", C.c_float)]


class SndParams(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("win_samples", C.c_int), ("step_samples", C.c_int),
                ("stride_samples", C.c_int), ("segment_steps", C.c_int),
                ("border_steps", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_msec_to_samples.restype = C.c_int
        L.orc_msec_to_samples.argtypes = [C.c_double, C.c_int]
        L.orc_seg_cnt.restype = C.c_int
        L.orc_seg_cnt.argtypes = [C.c_int] * 4
        L.orc_pcm_to_float.restype = C.c_double
        L.orc_pcm_to_float.argtypes = [C.c_int, C.c_int]
        L.orc_fft_plan_create.restype = C.c_void_p
        L.orc_fft_plan_create.argtypes = [C.c_int]
        L.orc_fft_plan_destroy.argtypes = [C.c_void_p]
        L.orc_fft_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_dft_naive_ld.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_freq_to_mel.restype = C.c_double
        L.orc_freq_to_mel.argtypes = [C.c_double]
        L.orc_mel_to_freq.restype = C.c_double
        L.orc_mel_to_freq.argtypes = [C.c_double]
        L.orc_freq_to_bin.restype = C.c_int
        L.orc_freq_to_bin.argtypes = [C.c_double] * 3
        L.orc_mel_init_filters.restype = C.c_int
        L.orc_dft_filter.restype = C.c_int
        L.orc_mel_filter_dft.restype = C.c_int
        L.orc_gabor_convolve.restype = C.c_int
        L.orc_process_segment.restype = C.c_int
        L.orc_process_batch.restype = C.c_int
        L.orc_snd_to_window.restype = C.c_int
        L.orc_mfcc_tail.restype = C.c_int
        L.orc_process_segment_mfcc.restype = C.c_int
        L.orc_fast_exp.restype = C.c_float
        L.orc_fast_exp.argtypes = [C.c_float]
        L.orc_noisy_xx1.restype = C.c_float
        L.orc_noisy_xx1.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.orc_kwta_layer.restype = C.c_int
        L.orc_kwta_pool.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def msec_to_samples(ms, rate):
    return lib().orc_msec_to_samples(float(ms), int(rate))


def sound_params(win_ms=25.0, step_ms=10.0, segment_ms=100.0, stride_ms=100.0, border_steps=2,
                 sr=16000):
    """sound/sndenv.go:202-207 -> SndParams (+ segment_samples as attribute)"""
    vals = [C.c_int() for _ in range(5)]
    lib().orc_sound_params(C.c_double(win_ms), C.c_double(step_ms), C.c_double(segment_ms),
                           C.c_double(stride_ms), C.c_int(border_steps), C.c_int(sr),
                           *[C.byref(v) for v in vals])
    sp = SndParams(sr, vals[0].value, vals[1].value, vals[3].value, vals[4].value, border_steps)
    sp.segment_samples = vals[2].value
    return sp


def dft_defaults():
    d = DftParams()
    lib().orc_dft_defaults(C.byref(d))
    return d


def mel_defaults():
    m = MelFBank()
    lib().orc_mel_defaults(C.byref(m))
    return m


def fft(x):
    """forward unnormalised DFT of a complex128 vector (own mixed-radix FFT)"""
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    pl = lib().orc_fft_plan_create(len(x))
    lib().orc_fft_forward(C.c_void_p(pl), _p(x), _p(out))
    lib().orc_fft_plan_destroy(C.c_void_p(pl))
    return out


def dft_naive_ld(x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    lib().orc_dft_naive_ld(_p(x), _p(out), len(x))
    return out


def mel_init_filters(m, dft_size, sample_rate):
    """mel.go:77-117. Returns (rc, bin_pts int32[nf+2], hz_pts, filters f64[nf, nf+2]); mutates m.renorm"""
    nf = m.n_filters
    bin_pts = np.zeros(nf + 2, np.int32)
    hz = np.zeros(nf + 2, np.float64)
    filt = np.zeros((nf, nf + 2), np.float64)
    rc = lib().orc_mel_init_filters(C.byref(m), C.c_int(dft_size), C.c_int(sample_rate),
                                    _p(bin_pts), _p(hz), _p(filt))
    return rc, bin_pts, hz, filt


def dft_filter(d, step, window, power, log_power, power_seg, log_power_seg, faithful=False):
    """dft.go:42-85; arrays are modified in place. power_seg: [H, T]"""
    n = len(window)
    T = power_seg.shape[1]
    window = np.ascontiguousarray(window, np.float64)
    pl = None if faithful else lib().orc_fft_plan_create(n)
    rc = lib().orc_dft_filter(C.byref(d), C.c_void_p(pl), C.c_int(step), _p(window), C.c_int(n),
                              _p(power), _p(log_power), _p(power_seg), _p(log_power_seg),
                              C.c_int(T))
    if pl:
        lib().orc_fft_plan_destroy(C.c_void_p(pl))
    return rc


def mel_filter_dft(m, bin_pts, step, power, segment, fbank, filters):
    """mel.go:120-153; segment [nf, T] and fbank [nf] modified in place"""
    return lib().orc_mel_filter_dft(C.byref(m), _p(bin_pts), C.c_int(step), _p(power),
                                    C.c_int(len(power)), _p(segment), C.c_int(segment.shape[1]),
                                    _p(fbank), _p(filters))


def gabor_specs(dicts):
    arr = (GaborSpec * len(dicts))()
    for i, d in enumerate(dicts):
        arr[i] = GaborSpec(int(d.get("off", 0)), d.get("wave_len", 0.0), d.get("orientation", 0.0),
                           d.get("sigma_width", 0.0), d.get("sigma_length", 0.0),
                           d.get("phase_offset", 0.0), int(d.get("circle_edge", 0)),
                           int(d.get("circular", 0)))
    return arr


def gabor_to_tensor(dicts, sx, sy, distribute=False):
    """gabor.go:89-222 on the active specs of `dicts` -> f64 [n, sy, sx]"""
    specs = gabor_specs(dicts)
    act = (GaborSpec * len(dicts))()
    n = lib().orc_gabor_active(specs, C.c_int(len(dicts)), act)
    out = np.zeros((n, sy, sx), np.float64)
    lib().orc_gabor_to_tensor(act, C.c_int(n), C.c_int(sx), C.c_int(sy), C.c_int(int(distribute)),
                              _p(out))
    return out


def gabor_convolve(mel, k, stride_x, stride_y, gain, out, by_time=False):
    """gabor.go:225-315; `out` (float32, rank 2 or 4) is modified in place; returns rc"""
    mel = np.ascontiguousarray(mel, np.float64)
    k = np.ascontiguousarray(k, np.float64)
    assert out.dtype == np.float32 and out.flags.c_contiguous
    shape = (C.c_int * out.ndim)(*out.shape)
    n_g, sy, sx = k.shape
    return lib().orc_gabor_convolve(_p(mel), C.c_int(mel.shape[0]), C.c_int(mel.shape[1]), _p(k),
                                    C.c_int(n_g), C.c_int(sx), C.c_int(sy), C.c_int(stride_x),
                                    C.c_int(stride_y), C.c_double(gain), _p(out),
                                    C.c_int(out.ndim), shape, C.c_int(int(by_time)))


def process_segment(sp, d, m, bin_pts, filters, signal, segment=0, add_ms=0, faithful=False):
    """sndenv.go:342-359 loop part. Returns dict of float64 arrays + frames done"""
    N, T, nf = sp.win_samples, sp.segment_steps, m.n_filters
    H = N // 2 + 1
    signal = np.ascontiguousarray(signal, np.float64)
    power = np.zeros(H)
    log_power = np.zeros(H)
    power_seg = np.zeros((H, T))
    log_power_seg = np.zeros((H, T))
    mel_seg = np.zeros((nf, T))
    fbank = np.zeros(nf)
    done = lib().orc_process_segment(C.byref(sp), C.byref(d), C.byref(m), _p(bin_pts),
                                     _p(filters), _p(signal), C.c_long(len(signal)),
                                     C.c_int(segment), C.c_int(add_ms), C.c_int(int(faithful)),
                                     _p(power), _p(log_power), _p(power_seg), _p(log_power_seg),
                                     _p(mel_seg), _p(fbank))
    return dict(done=done, power=power, log_power=log_power, power_seg=power_seg,
                log_power_seg=log_power_seg, mel_seg=mel_seg, fbank=fbank)


def process_batch(sp, d, m, bin_pts, filters, sig, sig_off, sig_len, seg, faithful=False,
                  gabor=None):
    """n_items x (process_segment [+ 4-D Convolve]).
    gabor = dict(k=f64[nG,sy,sx], stride_x, stride_y, gain, py, px) or None.
    Returns (rc, mel f64 [n, nf, T], gabor f32 [n, py, px, 2, nG] or None)"""
    n = len(sig_off)
    nf, T = m.n_filters, sp.segment_steps
    sig = np.ascontiguousarray(sig, np.float64)
    sig_off = np.ascontiguousarray(sig_off, np.int64)
    sig_len = np.ascontiguousarray(sig_len, np.int32)
    seg = np.ascontiguousarray(seg, np.int32)
    mel = np.zeros((n, nf, T), np.float64)
    if gabor is not None:
        k = np.ascontiguousarray(gabor["k"], np.float64)
        n_g, sy, sx = k.shape
        gout = np.zeros((n, gabor["py"], gabor["px"], 2, n_g), np.float32)
        gargs = [_p(k), C.c_int(n_g), C.c_int(sx), C.c_int(sy), C.c_int(gabor["stride_x"]),
                 C.c_int(gabor["stride_y"]), C.c_double(gabor["gain"]), C.c_int(gabor["py"]),
                 C.c_int(gabor["px"]), _p(gout)]
    else:
        gout = None
        gargs = [C.c_void_p(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0),
                 C.c_double(0), C.c_int(0), C.c_int(0), C.c_void_p(0)]
    rc = lib().orc_process_batch(C.byref(sp), C.byref(d), C.byref(m), _p(bin_pts), _p(filters),
                                 _p(sig), _p(sig_off), _p(sig_len), _p(seg), C.c_int(n),
                                 C.c_int(int(faithful)), _p(mel), *gargs)
    return rc, mel, gout


def dct1(x):
    """gonum fourier.DCT.Transform (FFTPACK cost): unnormalised DCT-I"""
    x = np.ascontiguousarray(x, np.float64)
    y = np.zeros_like(x)
    lib().orc_dct1(_p(x), _p(y), C.c_int(len(x)))
    return y


def process_segment_mfcc(sp, d, m, bin_pts, filters, signal, segment=0, add_ms=0, n_coefs=13, deltas=True):
    """sndenv.go:342-432 with Mel.MFCC on: mel + MFCCSegment / Energy / deltas / delta-deltas"""
    N, T, nf = sp.win_samples, sp.segment_steps, m.n_filters
    H = N // 2 + 1
    signal = np.ascontiguousarray(signal, np.float64)
    power, log_power = np.zeros(H), np.zeros(H)
    power_seg, log_power_seg = np.zeros((H, T)), np.zeros((H, T))
    mel_seg, fbank = np.zeros((nf, T)), np.zeros(nf)
    energy = np.zeros(T)
    mfcc, dl, ddl = np.zeros((n_coefs, T)), np.zeros((n_coefs, T)), np.zeros((n_coefs, T))
    done = lib().orc_process_segment_mfcc(C.byref(sp), C.byref(d), C.byref(m), _p(bin_pts), _p(filters),
                                          _p(signal), C.c_long(len(signal)), C.c_int(segment),
                                          C.c_int(add_ms), _p(power), _p(log_power), _p(power_seg),
                                          _p(log_power_seg), _p(mel_seg), _p(fbank), C.c_int(n_coefs),
                                          C.c_int(int(deltas)), _p(energy), _p(mfcc), _p(dl), _p(ddl))
    return dict(done=done, mel_seg=mel_seg, log_power_seg=log_power_seg, power_seg=power_seg, energy=energy,
                mfcc=mfcc, deltas=dl, delta_deltas=ddl)


# ---- k-WTA stage (kwta_oracle.c; third-party algorithm restated from memory of the published source) ----

def kwta_defaults():
    k = Kwta()
    lib().orc_kwta_defaults(C.byref(k))
    return k


def kwta_update(k):
    d = KwtaDerived()
    lib().orc_kwta_update(C.byref(k), C.byref(d))
    return d


def fast_exp(x):
    return lib().orc_fast_exp(float(x))


def noisy_xx1(k, x):
    d = kwta_update(k)
    return lib().orc_noisy_xx1(C.byref(k), C.byref(d), float(x))


def kwta_layer(k, raw):
    """KWTALayer on one tensor: returns (settled activations, cycles run); starts from act = raw"""
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    act = raw.copy()
    cy = lib().orc_kwta_layer(C.byref(k), _p(raw), _p(act), C.c_int(raw.size))
    return act, cy


def kwta_pool(k, raw, state=None):
    """KWTAPool on one [d0, d1, d2, d3] tensor: returns (act, cycles); state float32 [d0*d1, 2] is
    updated in place when given (the carried fffb.Inhibs slice)"""
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    assert raw.ndim == 4
    act = raw.copy()
    if state is not None:
        assert state.dtype == np.float32 and state.shape == (raw.shape[0] * raw.shape[1], 2) \
            and state.flags.c_contiguous
    cy = lib().orc_kwta_pool(C.byref(k), _p(raw), _p(act), *[C.c_int(v) for v in raw.shape],
                             _p(state) if state is not None else None)
    return act, cy
