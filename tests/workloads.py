"""Shared builders for tests / bench: the BASELINE configurations as (oracle params, product plan)."""
import numpy as np

from auditory_amd import capi, synth

DEFAULT_GABOR_SPECS = [dict(wave_len=2.0, orientation=o, sigma_width=0.5, sigma_length=0.5,
                            phase_offset=ph, circle_edge=1)
                       for o in (0, 45, 90, 135) for ph in (0, 1.5708)]   # processspeech.go:236-252

# name: sr, win_ms, step_ms, segment_ms, stride_ms, border, nf, lo, hi
CONFIGS = {
    "sndenv_16k_n400_nf32": (16000, 25.0, 10.0, 100.0, 100.0, 2, 32, 0.0, 8000.0),
    "cfg2_16k_n400_nf40": (16000, 25.0, 10.0, 1000.0, 1000.0, 2, 40, 0.0, 8000.0),
    "cfg2_16k_n512_nf40": (16000, 32.0, 10.0, 1000.0, 1000.0, 2, 40, 0.0, 8000.0),
    "cfg1_44k_n1103_nf32": (44100, 25.0, 10.0, 100.0, 100.0, 2, 32, 0.0, 8000.0),
    "cfg5_44k_n2048_nf128": (44100, 46.44, 10.0, 5000.0, 5000.0, 2, 128, 0.0, 22050.0),
    "odd_15k_n375_nf32": (15000, 25.0, 10.0, 100.0, 100.0, 2, 32, 0.0, 7000.0),     # 375 = 3 * 5^3
    "mixed_16k_n480_nf32": (16000, 30.0, 10.0, 200.0, 200.0, 3, 32, 300.0, 8000.0),  # 480 = 2^5 * 3 * 5
    # many narrow filters: the filter-group schedule outgrows one / two entries per thread of the staging code
    "many_16k_n512_nf124": (16000, 32.0, 10.0, 100.0, 100.0, 2, 124, 0.0, 8000.0),
    "many_16k_n400_nf64": (16000, 25.0, 10.0, 100.0, 100.0, 2, 64, 0.0, 8000.0),
    # processspeech's parameters at other common sample rates (tools/rate_sweep.py): smooth window lengths on the any-N kernel
    "rate_8k_n200_nf32": (8000, 25.0, 10.0, 100.0, 100.0, 2, 32, 0.0, 4000.0),       # 200 = 2^3 * 5^2
    "rate_22k_n551_nf32": (22050, 25.0, 10.0, 100.0, 100.0, 2, 32, 0.0, 8000.0),     # 551 = 19 * 29: Bluestein, L = 1152
    "rate_24k_n600_nf32": (24000, 25.0, 10.0, 100.0, 100.0, 2, 32, 0.0, 8000.0),     # 600 = 2^3 * 3 * 5^2
    "rate_32k_n800_nf32": (32000, 25.0, 10.0, 100.0, 100.0, 2, 32, 0.0, 8000.0),     # 800 = 2^5 * 5^2
    "rate_48k_n1200_nf32": (48000, 25.0, 10.0, 100.0, 100.0, 2, 32, 0.0, 8000.0),    # 1200 = 2^4 * 3 * 5^2
    "rate_96k_n2400_nf32": (96000, 25.0, 10.0, 100.0, 100.0, 2, 32, 0.0, 8000.0),    # 2400 = 2^5 * 3 * 5^2
    # 44.1 kHz = 2^2 3^2 5^2 7^2 Hz: windows other than the 25 ms default carry factors of 7 (radix-7 stages)
    "win20_44k_n882_nf32": (44100, 20.0, 10.0, 100.0, 100.0, 2, 32, 0.0, 8000.0),    # 882 = 2 * 3^2 * 7^2
    "win50_44k_n2205_nf64": (44100, 50.0, 10.0, 100.0, 100.0, 2, 64, 0.0, 8000.0),   # 2205 = 3^2 * 5 * 7^2 (odd: M = N)
}


class OracleCfg:
    """oracle-side parameter blocks + tables for one configuration"""

    def __init__(self, orc, name, segment_ms=None):
        sr, win, step, seg, stride, border, nf, lo, hi = CONFIGS[name]
        if segment_ms is not None:
            seg = stride = segment_ms
        self.name, self.sr, self.border = name, sr, border
        self.sp = orc.sound_params(win, step, seg, stride, border, sr)
        self.d = orc.dft_defaults()
        self.m = orc.mel_defaults()
        self.m.n_filters, self.m.lo_hz, self.m.hi_hz = nf, lo, hi
        rc, self.bins, self.hz, self.filt = orc.mel_init_filters(self.m, self.sp.win_samples, sr)
        assert rc == 0
        self.N, self.S, self.T = self.sp.win_samples, self.sp.step_samples, self.sp.segment_steps
        self.H, self.nf = self.N // 2 + 1, nf

    def full_len(self):
        """samples so that every frame of segment 0 is in bounds"""
        return self.S * (self.T - 1 - self.border) + self.N


def product_plan(ocfg, compute_dtype=capi.AUD_F64, gabor=None, device=0, dft_override=None, mfcc_coefs=0,  # noqa: C901
                 dft_log_offset=None, mel_log_off=None, mel_renorm_scale=None):
    """runtime.Plan built from the PRODUCT's own host setup for the same configuration"""
    from auditory_amd import agabor, capi, mel, runtime
    sr, win, step, seg, stride, border, nf, lo, hi = CONFIGS[ocfg.name]
    mp = mel.Params()
    mp.Defaults()
    mp.FBank.NFilters, mp.FBank.LoHz, mp.FBank.HiHz = nf, lo, hi
    filt = mp.InitFilters(ocfg.N, sr)
    if mel_log_off is not None:
        mp.FBank.LogOff = mel_log_off
    if mel_renorm_scale is not None:            # the user re-enables Renorm after InitFilters forced it off (Q5)
        mp.FBank.Renorm, mp.FBank.RenormScale = True, mel_renorm_scale
    dftp = capi.DftParams()
    capi.load().aud_dft_defaults(dftp)
    if dft_override is not None:
        dftp.prev_smooth, dftp.cur_smooth = dft_override
    if dft_log_offset is not None:
        dftp.log_offset = dft_log_offset
    gset = gk = None
    if gabor is not None:
        fs = agabor.FilterSet()
        fs.SizeX, fs.SizeY, fs.StrideX, fs.StrideY, fs.Gain = gabor["size"] + gabor["stride"] + (gabor["gain"],)
        specs = [agabor.Filter(WaveLen=s["wave_len"], Orientation=s["orientation"],
                               SigmaWidth=s["sigma_width"], SigmaLength=s["sigma_length"],
                               PhaseOffset=s["phase_offset"], CircleEdge=bool(s["circle_edge"]))
                 for s in gabor["specs"]]
        agabor.ToTensor(specs, fs)
        gset, gk = fs.to_c(), fs.Filters
    return runtime.Plan(runtime.get_ctx(device), ocfg.N, ocfg.S, ocfg.T, ocfg.border, dftp,
                        mp.FBank.to_c(), mp.BinPts, filt, gset, gk, compute_dtype, mfcc_coefs=mfcc_coefs)


def close_enough(got, ref, tol):
    """|got - ref| <= tol * max(1, |ref|), NaNs must coincide"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    if not np.array_equal(nan_g, nan_r):
        return False, "NaN pattern differs"
    ok = ~nan_r
    err = np.abs(got[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok]))
    worst = float(err.max()) if err.size else 0.0
    return worst <= tol, "max scaled err %.3g (tol %.1g)" % (worst, tol)


def spectrum_close(got, ref, tol, log_offset=None):
    """Per-bin power (or log-power, if log_offset is given: ln(p + offset)) against the oracle.
    A float32 FFT carries an absolute error proportional to the frame's LARGEST bin, so bins are
    judged relative to their frame's peak power: |dp| <= tol * max_k p_ref[k].  Arrays are
    [items, H, T]."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if log_offset is not None:
        dead = (ref == 0)                       # masked frames store 0, not ln(offset)
        got = np.where(dead, 0.0, np.exp(got) - log_offset)
        ref = np.where(dead, 0.0, np.exp(ref) - log_offset)
    peak = np.maximum(ref.max(axis=1, keepdims=True), 1.0)
    worst = float((np.abs(got - ref) / peak).max()) if got.size else 0.0
    return worst <= tol, "max |dp| / frame peak %.3g (tol %.1g)" % (worst, tol)


# ---- the parity criterion (BASELINE.json north_star: "within 1e-5 relative on the float32 mel /
# gabor tensors") -------------------------------------------------------------------------------
TOL = 1e-5          # |got - ref| <= TOL * max(1, |ref|)
TOL_F64 = 3e-7      # float64 compute: only the float32 rounding of the stored result is left
TAIL_FRAC = 5e-4    # f32 compute: share of elements allowed past TOL (at least 2) ...
TAIL_TOL = 2e-4     # ... and the bound those must still meet


# float64 plans, the gabor tensor: built on the float32-STORED mel values (81 of them per sum).  Their default kernel (k_gabor:
# float64 taps, float64 multiply-adds, as gabor.go:268-283) stays within 4e-7 of the oracle's Convolve on ITS float64 mel and is
# checked at TOL_F64_GABOR; the LDS-staged kernel a float64 plan can opt into (option gabor_kernel = 0: float32 taps, float32 row
# sums added in float64, gabor_tile.h) measured <= 1.7e-6 and is checked at TOL_F64_DERIVED
# (parity_cases.case_gabor_4d_and_2d_vs_oracle, case_process_fused_vs_oracle).  The north star's criterion for this tensor is 1e-5.
TOL_F64_GABOR = 1e-6
TOL_F64_DERIVED = 2.5e-6


def feature_close(got, ref, compute_dtype, lin_axis=None, tol_f64=None):
    """Compare a log-domain feature tensor (mel, or gabor built on it) with the oracle.

    float64 compute: every element within TOL_F64 (scaled by max(1, |ref|)).

    float32 compute: every element within TOL except for a tail of at most TAIL_FRAC of the
    elements (at least two), which must stay within TAIL_TOL.  Why a tail: a mel value is the log
    of a band power, and the narrow low-frequency triangles (weights 0,1,0) are the log of ONE
    bin.  A float32 FFT -- any float32 FFT, the input rounding alone does it -- leaves an absolute
    error of ~1e-7 x the frame's rms spectrum in every bin, so a bin that happens to sit 20 dB
    under its neighbours (Rayleigh-distributed noise does that in ~1 % of the bins) gets a
    relative error 10-100x the typical 3e-7, and its log moves by that much.  With lin_axis set
    the same data is also checked in the linear domain, where float32 is uniformly accurate:
    |exp(got) - exp(ref)| <= 4e-6 * max over lin_axis of exp(ref)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if got.shape != ref.shape:
        return False, "shape %s vs %s" % (got.shape, ref.shape)
    if not np.array_equal(np.isnan(got), np.isnan(ref)):
        return False, "NaN pattern differs"
    ok = ~np.isnan(ref)
    err = np.where(ok, np.abs(np.where(ok, got, 0) - np.where(ok, ref, 0)) / np.maximum(1.0, np.abs(np.where(ok, ref, 0))), 0.0)
    worst = float(err.max()) if err.size else 0.0
    if compute_dtype == capi.AUD_F64:
        t64 = tol_f64 if tol_f64 is not None else TOL_F64
        return worst <= t64, "max scaled err %.3g (tol %.1g, f64)" % (worst, t64)
    n_out = int((err > TOL).sum())
    allowed = max(2, int(np.ceil(TAIL_FRAC * err.size)))
    msg = "RELAXED float32 criterion: max scaled err %.3g, %d of %d past %.0e (allowed %d, tail bound %.0e)" % (
        worst, n_out, err.size, TOL, allowed, TAIL_TOL)
    good = n_out <= allowed and worst <= TAIL_TOL
    if good and lin_axis is not None:
        eg, er = np.exp(np.where(ok, got, -np.inf)), np.exp(np.where(ok, ref, -np.inf))
        peak = np.maximum(er.max(axis=lin_axis, keepdims=True), 1e-30)
        lin = float((np.abs(eg - er) / peak).max())
        msg += "; linear-domain %.3g" % lin
        good = lin <= 4e-6
    return good, msg
