"""Pins the CPU oracle (oracle/auditory_oracle.c): first-principles KATs and an
independent numpy/long-double cross-check.  The reference itself has no tests or
golden vectors -- "parity unpinned" -- so this is what anchors it."""
import os

import numpy as np
import pytest

import np_ref
import np_ref as R
from auditory_amd import synth


# ---------------- DFT ------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8, 12, 25, 30, 64, 97, 200, 400, 512, 1103])
def test_fft_matches_numpy_and_longdouble(orc, n):
    rng = np.random.default_rng(n)
    x = rng.normal(size=n) + 1j * rng.normal(size=n)
    got = orc.fft(x)
    ref = np.fft.fft(x)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-13 * scale
    if n <= 400:
        ld = orc.dft_naive_ld(x)
        assert np.abs(got - ld).max() <= 2e-13 * scale


def test_fft_kats(orc):
    n = 400
    imp = np.zeros(n, complex); imp[0] = 1
    assert np.allclose(orc.fft(imp), 1.0, atol=1e-15)
    dc = np.ones(n, complex)
    X = orc.fft(dc)
    assert abs(X[0] - n) < 1e-12 and np.abs(X[1:]).max() < 1e-11
    # amplitude-A tone at an exact bin: |X[k]|^2 = (A n / 2)^2
    A, k = 0.8, 37
    tone = A * np.cos(2 * np.pi * k * np.arange(n) / n)
    P = np.abs(orc.fft(tone)) ** 2
    assert abs(P[k] - (A * n / 2) ** 2) < 1e-8
    H = n // 2 + 1
    assert P[:H][np.arange(H) != k].max() < 1e-24 * (A * n / 2) ** 2 + 1e-20


# ---------------- params ---------------------------------------------------
def test_msec_to_samples(orc):
    assert orc.msec_to_samples(25, 16000) == 400
    assert orc.msec_to_samples(32, 16000) == 512
    assert orc.msec_to_samples(25, 44100) == 1103      # 1102.5 rounds away from zero
    assert orc.msec_to_samples(10, 44100) == 441
    assert orc.msec_to_samples(46.44, 44100) == 2048
    sp = orc.sound_params(25, 10, 100, 100, 2, 16000)
    assert (sp.win_samples, sp.step_samples, sp.segment_steps, sp.stride_samples) == (400, 160, 14, 1600)
    sp = orc.sound_params(32, 10, 1000, 1000, 2, 16000)
    assert (sp.win_samples, sp.segment_steps) == (512, 104)
    assert orc.lib().orc_seg_cnt(48000, 1600, 1600, 1) == 30
    assert orc.lib().orc_pcm_to_float(32767, 16) == 1.0
    assert orc.lib().orc_pcm_to_float(-32768, 16) == -32768 / 32767


# ---------------- mel ------------------------------------------------------
@pytest.mark.parametrize("nf,n,sr,hi", [(32, 400, 16000, 8000), (40, 400, 16000, 8000),
                                        (40, 512, 16000, 8000), (32, 1103, 44100, 8000),
                                        (128, 2048, 44100, 22050)])
def test_mel_table_invariants(orc, nf, n, sr, hi):
    m = orc.mel_defaults()
    m.n_filters, m.hi_hz = nf, hi
    assert m.renorm == 1
    rc, bins, hz, filt = orc.mel_init_filters(m, n, sr)
    assert rc == 0
    assert m.renorm == 0                       # mel.go:80 forces it off
    nb, nhz = np_ref.mel_table(nf, 0.0, hi, n, sr)
    assert np.array_equal(bins, nb) and np.allclose(hz, nhz, rtol=1e-14)
    assert np.all(np.diff(bins) >= 0) and bins[0] == 0
    assert bins[-1] <= n // 2                  # FilterDft never reads past Power
    widths = bins[2:] - bins[:-2] + 1
    assert widths.max() <= nf + 2              # Q4 envelope holds for every BASELINE config
    for f in range(nf):
        lo, c, hi_ = bins[f], bins[f + 1], bins[f + 2]
        row = filt[f, :hi_ - lo + 1]
        if c == lo:                            # degenerate -> NaN tap (Q3)
            assert np.isnan(row[0])
            continue
        assert row[0] == 0 and row[c - lo] == 1.0
        if hi_ > c:
            assert row[-1] == 0
        assert np.all(np.diff(row[:c - lo + 1]) > 0) and np.all(np.diff(row[c - lo:]) < 0)
    if (nf, n) == (128, 2048):
        assert np.isnan(filt[0, 0])            # the NaN row of cfg 5


def test_mel_table_spill_and_panic(orc):
    # a triangle wider than nf+2 spills into the next row (Q4); past the end => "panic"
    m = orc.mel_defaults()
    m.n_filters, m.hi_hz = 4, 8000
    rc, bins, hz, filt = orc.mel_init_filters(m, 512, 16000)
    assert rc == orc.ORC_EPANIC


# ---------------- segment loop vs numpy ------------------------------------
CFGS = [  # name, sr, win_ms, seg_ms, nf, hi, dur_s
    ("16k_n400_nf32_seg100", 16000, 25.0, 100.0, 32, 8000.0, 0.5),
    ("16k_n400_nf40_1s", 16000, 25.0, 1000.0, 40, 8000.0, 1.0),
    ("16k_n512_nf40_1s", 16000, 32.0, 1000.0, 40, 8000.0, 1.0),
    ("44k_n1103_nf32", 44100, 25.0, 100.0, 32, 8000.0, 0.3),
    ("44k_n2048_nf128", 44100, 46.44, 200.0, 128, 22050.0, 0.4),
]


@pytest.mark.parametrize("cfg", CFGS, ids=[c[0] for c in CFGS])
def test_process_segment_vs_numpy(orc, cfg):
    name, sr, win_ms, seg_ms, nf, hi, dur = cfg
    sp = orc.sound_params(win_ms, 10.0, seg_ms, seg_ms, 2, sr)
    d = orc.dft_defaults()
    m = orc.mel_defaults()
    m.n_filters, m.hi_hz = nf, hi
    rc, bins, hz, filt = orc.mel_init_filters(m, sp.win_samples, sr)
    assert rc == 0
    sig, _ = synth.batch(1, 1, int(dur * sr), sr)
    sig = sig[0]
    for seg in (0, 1):
        r = orc.process_segment(sp, d, m, bins, filt, sig, segment=seg)
        mel, P, lp = np_ref.melspec(sig, sp.win_samples, sp.step_samples, sp.segment_steps, 2, nf,
                                    0.0, hi, sr, seg_start=seg * sp.stride_samples)
        assert r["done"] > 0 or seg == 1
        np.testing.assert_allclose(r["power_seg"], P, rtol=1e-9, atol=1e-12 * P.max())
        np.testing.assert_allclose(r["log_power_seg"], lp, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r["mel_seg"], mel, rtol=1e-10, atol=1e-11, equal_nan=True)
        if name == "44k_n2048_nf128":
            assert np.isnan(r["mel_seg"][0, :r["done"]]).all()     # Q3
            assert not np.isnan(r["mel_seg"][1:]).any()


def test_segment_quirks(orc):
    sp = orc.sound_params(25, 10, 100, 100, 2, 16000)
    d, m = orc.dft_defaults(), orc.mel_defaults()
    rc, bins, hz, filt = orc.mel_init_filters(m, 400, 16000)
    sig, _ = synth.batch(2, 1, 1900, 16000)
    sig = sig[0]
    r = orc.process_segment(sp, d, m, bins, filt, sig, segment=0)
    # frames 0,1 start at -320,-160 (partly zero-padded), all 14 fit? last start = 160*11, end 2160 > 1900
    assert r["done"] == 12                                       # Q7: loop breaks at first short frame
    assert np.all(r["mel_seg"][:, 12:] == 0) and np.all(r["power_seg"][:, 12:] == 0)
    assert np.all(r["log_power_seg"][:, 12:] == 0)
    # all-zero signal => power 0 => logpower ln(1)=0, mel = LogMin = -10 exactly (Q2)
    z = orc.process_segment(sp, d, m, bins, filt, np.zeros(4000), segment=0)
    assert z["done"] == 14
    assert np.all(z["mel_seg"] == -10.0) and np.all(z["log_power_seg"] == 0.0)
    # add (ms) shifts the start, sndenv.go:440
    a = orc.process_segment(sp, d, m, bins, filt, sig, segment=0, add_ms=10)
    assert np.allclose(a["mel_seg"][:, 0], r["mel_seg"][:, 1])


def test_prev_smooth_recurrence(orc):
    # dft.go:67-69: p_s = Prev*p_{s-1} + Cur*raw_s for s>0; p_0 = raw_0 (Q6)
    sp = orc.sound_params(25, 10, 100, 100, 2, 16000)
    m = orc.mel_defaults()
    rc, bins, hz, filt = orc.mel_init_filters(m, 400, 16000)
    sig, _ = synth.batch(3, 1, 8000, 16000)
    d0 = orc.dft_defaults()
    raw = orc.process_segment(sp, d0, m, bins, filt, sig[0], segment=1)["power_seg"]
    d = orc.dft_defaults()
    d.prev_smooth, d.cur_smooth = 0.3, 0.7
    sm = orc.process_segment(sp, d, m, bins, filt, sig[0], segment=1)["power_seg"]
    exp = raw.copy()
    for s in range(1, raw.shape[1]):
        exp[:, s] = 0.3 * exp[:, s - 1] + 0.7 * raw[:, s]
    np.testing.assert_allclose(sm, exp, rtol=1e-13)


def test_mel_renorm_branch(orc):
    m = orc.mel_defaults()
    rc, bins, hz, filt = orc.mel_init_filters(m, 400, 16000)
    m.renorm, m.renorm_scale = 1, 1.0 / (m.renorm_max - m.renorm_min)   # user re-enables after Init (Q5)
    P = np.abs(np.fft.fft(np.random.default_rng(0).normal(size=400))[:201]) ** 2
    seg = np.zeros((32, 1)); fb = np.zeros(32)
    assert orc.mel_filter_dft(m, bins, 0, P, seg, fb, filt) == 0
    W = np_ref.mel_weights_dense(bins, 32, 201)
    exp = np.clip((np.log(W @ P) + 6.0), 0, None) * 0.1
    exp = np.minimum(exp, 1.0)
    np.testing.assert_allclose(fb, exp, rtol=1e-12)


# ---------------- gabor ----------------------------------------------------
DEFAULT_SPECS = [dict(wave_len=2.0, orientation=o, sigma_width=0.5, sigma_length=0.5,
                      phase_offset=ph, circle_edge=1)
                 for o in (0, 45, 90, 135) for ph in (0, 1.5708)]     # processspeech.go:236-252


def test_gabor_kernel_invariants(orc):
    k = orc.gabor_to_tensor(DEFAULT_SPECS, 9, 9)
    assert k.shape == (8, 9, 9)
    for g in range(8):
        assert abs(k[g][k[g] > 0].sum() - 1.0) < 1e-12               # gabor.go:195-221
        assert abs(k[g][k[g] < 0].sum() + 1.0) < 1e-12
    assert np.all(k[:, 0, 0] == 0)                                   # CircleEdge corner
    # orientation-0 sine gabor is odd in y, constant sign along a row
    assert np.allclose(k[0], -k[0][::-1, :], atol=1e-15)
    # Off specs are dropped (gabor.go:329-336); zero fields get defaults (gabor.go:73-86)
    specs = [dict(off=1, wave_len=2.0, orientation=0, sigma_width=0.5, sigma_length=0.5, circle_edge=1),
             dict(orientation=45, circle_edge=1)]
    k2 = orc.gabor_to_tensor(specs, 9, 9)
    assert k2.shape == (1, 9, 9) and np.allclose(k2[0], k[2])
    # const input => fSum ~ 0 (pos + neg halves cancel)
    out = np.full((11, 32, 2, 8), 7.0, np.float32)
    assert orc.gabor_convolve(np.full((40, 104), 3.25), k, 3, 3, 2.0, out) == 0
    assert np.abs(out).max() < 1e-12 * 100


def test_gabor_distribute_and_circular(orc):
    specs = [dict(wave_len=2.0, orientation=0, sigma_width=0.5, sigma_length=0.5, circle_edge=1),
             dict(wave_len=2.0, orientation=0, sigma_width=0.5, sigma_length=0.5, circle_edge=1),
             dict(wave_len=1.5, sigma_width=0.4, circular=1)]
    k = orc.gabor_to_tensor(specs, 8, 8, distribute=True)
    assert k.shape == (3, 8, 8) and np.isfinite(k).all()
    assert not np.allclose(k[0], k[1])        # two 0-degree filters at different centres


def test_gabor_convolve_vs_numpy(orc):
    rng = np.random.default_rng(5)
    mel = rng.normal(3.0, 2.0, size=(40, 104))
    mel[0, :] = np.nan                                               # NaN -> 0.5 (gabor.go:278-280)
    k = orc.gabor_to_tensor(DEFAULT_SPECS, 9, 9)
    out = np.zeros((11, 32, 2, 8), np.float32)
    assert orc.gabor_convolve(mel, k, 3, 3, 2.0, out) == 0
    ref = np_ref.gabor4(mel, k, 3, 3, 2.0, 11, 32)
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-7)
    assert (out[:, :, 0, :] * out[:, :, 1, :] == 0).all()            # on/off are exclusive
    # 2-D modes (gaborview sizing gbv.go:800-813): 8x8, stride 6x3, 4 filters
    k4 = orc.gabor_to_tensor(DEFAULT_SPECS[::2], 8, 8)
    nfy, nfx = (40 - 8) // 3 + 1, (104 - 8) // 6 + 1
    o_bt = np.zeros((2 * nfy, nfx * 4), np.float32)
    o_bf = np.zeros_like(o_bt)
    assert orc.gabor_convolve(mel, k4, 6, 3, 1.5, o_bt, by_time=True) == 0
    assert orc.gabor_convolve(mel, k4, 6, 3, 1.5, o_bf, by_time=False) == 0
    for g in range(4):
        assert np.array_equal(o_bt[:, g * nfx:(g + 1) * nfx], o_bf[:, g::4])
    m2 = np.where(np.isnan(mel), 0.5, mel)
    s = (k4[1] * m2[3:11, 12:20]).sum()                              # fIdx=1, tIdx=2, flt=1
    assert abs(o_bf[2 + (0 if s >= 0 else 1), 1 + 2 * 4] - 1.5 * abs(s)) < 1e-5
    # rejects: mel narrower than filter; rank 5 output (processspeech no-op, Q9)
    o = np.ones((2, 2), np.float32)
    assert orc.gabor_convolve(mel[:, :5], k, 3, 3, 2.0, o) == orc.ORC_EINVAL and (o == 1).all()
    o5 = np.ones((1, 11, 32, 2, 8), np.float32)
    assert orc.gabor_convolve(mel, k, 3, 3, 2.0, o5) == orc.ORC_EINVAL and (o5 == 1).all()
    # over-large pools (Q10): reads run past the tensor => the Go code panics
    big = np.zeros((11, 40, 2, 8), np.float32)
    assert orc.gabor_convolve(mel, k, 3, 3, 2.0, big) in (orc.ORC_OK, orc.ORC_EPANIC)


def test_batch_driver(orc):
    sp = orc.sound_params(32, 10, 1000, 1000, 2, 16000)
    d, m = orc.dft_defaults(), orc.mel_defaults()
    m.n_filters = 40
    rc, bins, hz, filt = orc.mel_init_filters(m, 512, 16000)
    L = 160 * 101 + 512
    sig, _ = synth.batch(2, 3, 16000, 16000, row_len=L)
    k = orc.gabor_to_tensor(DEFAULT_SPECS, 9, 9)
    g = dict(k=k, stride_x=3, stride_y=3, gain=2.0, py=11, px=32)
    rc, mel, gout = orc.process_batch(sp, d, m, bins, filt, sig.ravel(), np.arange(3) * L,
                                      np.full(3, L), np.zeros(3), gabor=g)
    assert rc == 0 and mel.shape == (3, 40, 104) and gout.shape == (3, 11, 32, 2, 8)
    r1 = orc.process_segment(sp, d, m, bins, filt, sig[1])
    assert np.array_equal(mel[1], r1["mel_seg"])
    rc, mel_f, _ = orc.process_batch(sp, d, m, bins, filt, sig.ravel(), np.arange(3) * L,
                                     np.full(3, L), np.zeros(3), faithful=True)
    assert np.array_equal(mel, mel_f)


# ---------------- the reference's own WAV fixtures (only where present) ----
REF_SOUNDS = "/root/reference/examples/processspeech/sounds"


def _read_wav_mono16(path):
    import struct
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, sz = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", b[pos + 8:pos + 24])
        elif cid == b"data":
            data = b[pos + 8:pos + 8 + sz]
        pos += 8 + sz + (sz & 1)
    assert fmt[0] == 1 and fmt[1] == 1 and fmt[5] == 16
    return np.frombuffer(data, "<i2"), fmt[2]


@pytest.mark.skipif(not os.path.isdir(REF_SOUNDS), reason="reference WAV fixtures not on this box")
@pytest.mark.parametrize("hz,peak_bin", [(800, 20), (2000, 50), (5000, 125), (7000, 175)])
def test_reference_tone_fixtures(orc, hz, peak_bin):
    """The shipped pure-tone WAVs (44.1 kHz mono) through the processspeech
    parameters (N=1103 prime, nf=32): the DFT peak must sit at round(hz*N/sr) and
    the hottest mel row must be the filter whose triangle peaks nearest that bin."""
    pcm, sr = _read_wav_mono16(os.path.join(REF_SOUNDS, "%d.wav" % hz))
    assert sr == 44100
    sig = synth.pcm_to_float64(pcm)
    sp = orc.sound_params(25, 10, 100, 100, 2, sr)
    assert sp.win_samples == 1103
    d, m = orc.dft_defaults(), orc.mel_defaults()
    rc, bins, hzp, filt = orc.mel_init_filters(m, 1103, sr)
    r = orc.process_segment(sp, d, m, bins, filt, sig, segment=1)
    assert r["done"] == 14
    assert int(np.argmax(r["power_seg"][:, 5])) == peak_bin
    hot = int(np.argmax(r["mel_seg"][:, 5]))
    assert bins[hot] <= peak_bin <= bins[hot + 2]


# ---------------- MFCC tail -----------------------------------
def test_dct1_matches_scipy_and_fftpack_contract(orc):
    from scipy.fft import dct
    rng = np.random.default_rng(4)
    for n in (2, 3, 13, 32, 40):
        x = rng.normal(size=n)
        y = orc.dct1(x)
        assert np.abs(y - dct(x, type=1)).max() < 1e-12 * n       # unnormalised DCT-I
        assert np.abs(orc.dct1(y) - 2 * (n - 1) * x).max() < 1e-11 * n   # gonum: twice => x * 2(n-1)


def test_mfcc_tail_semantics(orc):
    sp = orc.sound_params(25, 10, 100, 100, 2, 16000)
    d, m = orc.dft_defaults(), orc.mel_defaults()
    rc, bins, hz, filt = orc.mel_init_filters(m, 400, 16000)
    sig, _ = synth.batch(8, 1, 4000, 16000)
    o = orc.process_segment_mfcc(sp, d, m, bins, filt, sig[0], segment=1)
    T, nc = 14, 13
    assert o["done"] == 14
    plain = orc.process_segment(sp, d, m, bins, filt, sig[0], segment=1)
    assert np.array_equal(o["mel_seg"], plain["mel_seg"])
    # Energy[s] = sum over the steps of the log-power of BIN s (the axis quirk, sndenv.go:360-366)
    np.testing.assert_allclose(o["energy"], o["log_power_seg"][:T, :].sum(axis=1), rtol=1e-13)
    assert np.array_equal(o["mfcc"][0], o["energy"])                     # row 0 overwritten (:368-372)
    # rows 1.. are the DCT-I of the log-mel column
    for s in (0, 5, 13):
        c = orc.dct1(o["mel_seg"][:, s])
        np.testing.assert_allclose(o["mfcc"][1:, s], c[1:nc], rtol=1e-12)
    # deltas: running sums carried across coefficients (sndenv.go:378-404), restated naively here
    M = o["mfcc"]
    exp = np.zeros_like(M)
    for s in range(T):
        prv = nxt = 0.0
        for i in range(nc):
            nume = 0.0
            for n in (1, 2):
                prv += M[i, max(s - n, 0)]
                nxt += M[i, min(s + n, T - 1)]
                nume += n * (nxt - prv)
                exp[i, s] = nume / (2 * n * n)
    np.testing.assert_allclose(o["deltas"], exp, rtol=1e-12, atol=1e-12)
    # a short signal: unprocessed steps keep MFCC rows 1.. at zero, row 0 still gets Energy
    o2 = orc.process_segment_mfcc(sp, d, m, bins, filt, sig[0][:1900], segment=0)
    assert o2["done"] == 12 and np.all(o2["mfcc"][1:, 12:] == 0)


# ---- k-WTA stage (oracle/kwta_oracle.c) ---------------------------------------------------------

def test_fast_exp_is_the_quartic_spline(orc):
    """FastExp tracks exp to ~1e-5 relative over the range NoisyXX1 uses it on (0, 50], is continuous across
    mantissa wrap-arounds, and agrees with an independent integer restatement bit for bit"""
    xs = np.linspace(-80.0, 80.0, 4001)
    rel = np.array([abs(orc.fast_exp(x) - np.exp(x)) / np.exp(x) for x in xs])
    assert rel.max() < 2e-5
    for x in (0.0, 1e-3, 0.37, 1.0, 17.25, 49.999, -3.5):
        assert np.float32(orc.fast_exp(x)) == R.fast_exp32(x)
    assert orc.fast_exp(-100.0) == 0.0


def test_noisy_xx1_shape(orc):
    k = orc.kwta_defaults()
    d = orc.kwta_update(k)
    assert abs(d.sig_gain_nvar - 300.0) < 1e-4 and abs(d.act_dt - 1 / 3) < 1e-7
    assert np.allclose(list(d.erev_sub_thr), [0.5, -0.2, -0.25, -0.4], atol=1e-7)
    xs = np.linspace(-0.3, 1.0, 5201).astype(np.float32)
    ys = np.array([orc.noisy_xx1(k, float(x)) for x in xs])
    assert ys[0] == 0.0 and (np.diff(ys) >= -2e-6).all() and 0.97 < ys[-1] < 1.0   # monotone (to FastExp's ripple), saturating
    # continuous where its three pieces meet (x = 0 and x = InterpRange)
    for x0 in (0.0, 0.01):
        lo, hi = orc.noisy_xx1(k, np.nextafter(np.float32(x0), np.float32(-1))), orc.noisy_xx1(k, x0)
        assert abs(hi - lo) < 1e-4
    assert abs(orc.noisy_xx1(k, 0.0) - d.sig_val_at0) < 1e-7
    ref = R.KwtaRef(k)
    for x in (-0.1, -0.01, -1e-4, 0.0, 0.004, 0.01, 0.03, 0.2, 0.9):
        assert np.float32(orc.noisy_xx1(k, x)) == ref.noisy_xx1(np.float32(x))


def test_kwta_pool_vs_python_restatement(orc):
    """the C oracle against a scalar numpy-float32 restatement written separately (small tensor)"""
    k = orc.kwta_defaults()
    rng = np.random.default_rng(4)
    raw = np.maximum(rng.normal(0, 0.4, (3, 4, 2, 4)), 0).astype(np.float32)
    act, cy = orc.kwta_pool(k, raw)
    ref, cy_ref = R.KwtaRef(k).pool(raw)
    assert cy == cy_ref and np.array_equal(act, ref)


def test_kwta_behaviour(orc):
    k = orc.kwta_defaults()
    rng = np.random.default_rng(8)
    raw = np.maximum(rng.normal(0, 0.35, (11, 32, 2, 8)), 0).astype(np.float32)
    act, cy = orc.kwta_pool(k, raw)
    assert 3 < cy <= k.iters and act.min() >= 0 and act.max() < 1
    on = act > 0.1
    assert 0.02 < on.mean() < 0.4                       # sparse
    assert raw[on].min() > np.median(raw)               # and it is the strong inputs that survive
    # more inhibition -> fewer active units
    k2 = orc.kwta_defaults()
    k2.pool.gi, k2.lay.gi = 3.0, 2.5
    act2, _ = orc.kwta_pool(k2, raw)
    assert (act2 > 0.1).sum() < on.sum()
    # no inhibition at all: every unit above the leak threshold fires
    k3 = orc.kwta_defaults()
    k3.pool.on = k3.lay.on = 0
    act3, _ = orc.kwta_pool(k3, raw)
    assert (act3 > 0.1).sum() > 2 * on.sum()
    # all-zero input: a fresh state stays (numerically) silent
    z, _ = orc.kwta_pool(k, np.zeros((2, 3, 2, 8), np.float32))
    assert z.max() < 1e-3
    # pool level at work: the layer-only result differs
    lay, _ = orc.kwta_layer(k, raw)
    assert np.abs(lay - act).max() > 0.05
    # the carried state matters on the second call and converges
    st = np.zeros((11 * 32, 2), np.float32)
    a1, _ = orc.kwta_pool(k, raw, st)
    a2, _ = orc.kwta_pool(k, raw, st)
    assert np.array_equal(a1, act) and not np.array_equal(a2, a1)
