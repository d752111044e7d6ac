"""Shared builders for tests / bench: the BASELINE configurations as (oracle params, product plan)."""
import numpy as np

from auditory_amd import synth

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


def product_plan(ocfg, compute_dtype=0, gabor=None, device=0):  # noqa: C901
    """runtime.Plan built from the PRODUCT's own host setup for the same configuration"""
    from auditory_amd import agabor, capi, mel, runtime
    sr, win, step, seg, stride, border, nf, lo, hi = CONFIGS[ocfg.name]
    mp = mel.Params()
    mp.Defaults()
    mp.FBank.NFilters, mp.FBank.LoHz, mp.FBank.HiHz = nf, lo, hi
    filt = mp.InitFilters(ocfg.N, sr)
    dftp = capi.DftParams()
    capi.load().aud_dft_defaults(dftp)
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
                        mp.FBank.to_c(), mp.BinPts, filt, gset, gk, compute_dtype)


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
