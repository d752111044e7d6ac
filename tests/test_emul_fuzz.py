"""Seeded random parameter sets through the emulator build: window lengths with every kind of
factorisation (powers of two, 3/5-smooth, primes, prime x small), odd and even steps, borders,
segment lengths, filter counts, both compute types -- each against the oracle.  This is the
"any N" claim of the generic kernel (and the plan's kernel selection) under test."""
import os
import sys

import numpy as np
import pytest

import workloads as W
from auditory_amd import capi, mel as melmod, runtime, synth

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emul"))

WINDOWS = [4, 6, 8, 10, 16, 18, 25, 27, 30, 32, 49, 50, 64, 75, 96, 100, 121, 125, 128, 144, 150, 160, 169,
           200, 202, 211, 240, 243, 250, 256, 289, 300, 320, 343, 360, 375, 384, 400, 441, 450, 480, 500, 512,
           514, 600, 625, 640, 729, 750, 800, 882, 1000, 1024]


def _valid_mel(n_fft, sr, rng):
    """pick (nf, lo, hi) whose triangles fit the [nf, nf+2] table (the reference's own envelope)"""
    for _ in range(200):
        nf = int(rng.integers(3, 48))
        hi = float(rng.uniform(0.25, 0.5) * sr)
        lo = float(rng.uniform(0, 0.1) * sr)
        mp = melmod.Params()
        mp.Defaults()
        mp.FBank.NFilters, mp.FBank.LoHz, mp.FBank.HiHz = nf, lo, hi
        try:
            filt = mp.InitFilters(n_fft, sr)
        except capi.AuditoryError:
            continue
        if mp.BinPts[-1] <= n_fft // 2 and (np.diff(mp.BinPts) >= 0).all():
            return mp, filt
    return None, None


@pytest.mark.parametrize("seed", range(48))
def test_random_configs_vs_oracle(orc, seed):
    import backend
    rng = np.random.default_rng(1000 + seed)
    N = int(WINDOWS[int(rng.integers(0, len(WINDOWS)))])
    sr = int(rng.choice([8000, 11025, 16000, 22050]))
    S = int(rng.integers(max(1, N // 8), N + 3))
    T = int(rng.integers(1, 23))
    border = int(rng.integers(0, 4))
    cdt = capi.AUD_F64 if seed % 3 == 0 else capi.AUD_F32
    mp, filt = _valid_mel(N, sr, rng)
    if mp is None:
        pytest.skip("no mel table fits the reference's [nf, nf+2] envelope for N=%d" % N)
    nf = mp.FBank.NFilters
    L = int(rng.integers(N, N + S * (T + 2)))
    sig, _ = synth.batch(500 + seed, 2, L, sr)
    # oracle with the same numbers
    sp = orc.SndParams(sr, N, S, int(rng.integers(1, 3)) * S, T, border)
    d, m = orc.dft_defaults(), orc.mel_defaults()
    m.n_filters, m.lo_hz, m.hi_hz = nf, mp.FBank.LoHz, mp.FBank.HiHz
    rc, bins, hz, ofilt = orc.mel_init_filters(m, N, sr)
    assert rc == 0 and np.array_equal(bins, mp.BinPts)
    segs = [(r, s) for r in range(2) for s in (0, 1)]
    ref = [orc.process_segment(sp, d, m, bins, ofilt, sig[r], segment=s) for r, s in segs]
    dftp = capi.DftParams()
    capi.load().aud_dft_defaults(dftp)
    with backend.emulated("plain"):
        plan = runtime.Plan(runtime.get_ctx(0), N, S, T, border, dftp, mp.FBank.to_c(), mp.BinPts, filt,
                            compute_dtype=cdt)
        items = runtime.make_items([r * L for r, s in segs], [L] * len(segs), [s * sp.stride_samples for r, s in segs])
        mel, pw, lp = plan.melspec_host(sig.ravel(), items, True, True)
        plan.close()
    ref_mel = np.stack([o["mel_seg"] for o in ref])
    ref_pw = np.stack([o["power_seg"] for o in ref])
    ok, msg = W.feature_close(mel, ref_mel, cdt, lin_axis=1)
    assert ok, "N=%d S=%d T=%d border=%d nf=%d: mel %s" % (N, S, T, border, nf, msg)
    ok, msg = W.spectrum_close(pw, ref_pw, 4e-6 if cdt == capi.AUD_F32 else 3e-7)
    assert ok, "N=%d S=%d T=%d: power %s" % (N, S, T, msg)


@pytest.mark.parametrize("N", [400, 512, 2048])
@pytest.mark.parametrize("seed", range(8))
def test_random_wave_kernel_configs(orc, seed, N):
    import backend
    import parity_cases as PC
    with backend.emulated("plain"):
        print(PC.case_random_wave_config(orc, seed, N))


@pytest.mark.parametrize("seed", range(16))
def test_random_kwta_params_vs_oracle(orc, seed):
    """random k-WTA parameter sets and tensor shapes: every branch of FFFB (on/off, MaxVsAvg, FF0 above / below
    the average) and of NoisyXX1 (sigmoid tail, the exp cut-off, interpolation band, gain-corrected XX1 on both
    sides of GainCorRange) must come out bit-identical to the float32 oracle"""
    import backend
    import parity_cases as PC
    from auditory_amd import kwta
    rng = np.random.default_rng(7000 + seed)
    over = {
        "Iters": int(rng.integers(0, 25)), "DelActThr": float(rng.choice([0.0, 0.002, 0.005, 0.05])),
        "ActTau": float(rng.uniform(1.0, 6.0)),
        "LayFFFB.On": bool(rng.integers(0, 4)), "LayFFFB.Gi": float(rng.uniform(0.5, 3.0)),
        "LayFFFB.FF0": float(rng.uniform(0.0, 0.3)), "LayFFFB.MaxVsAvg": float(rng.choice([0.0, 0.0, 0.4])),
        "LayFFFB.FBTau": float(rng.uniform(1.0, 3.0)),
        "PoolFFFB.On": bool(rng.integers(0, 4)), "PoolFFFB.Gi": float(rng.uniform(0.5, 3.0)),
        "PoolFFFB.FF": float(rng.uniform(0.5, 1.5)), "PoolFFFB.FB": float(rng.uniform(0.0, 1.5)),
        "PoolFFFB.MaxVsAvg": float(rng.choice([0.0, 0.25, 1.0])),
        "XX1.Gain": float(rng.choice([20.0, 40.0, 80.0, 100.0, 600.0])), "XX1.NVar": float(rng.choice([0.005, 0.01, 0.02])),
        "XX1.GainCorRange": float(rng.choice([2.0, 10.0])), "XX1.Thr": float(rng.uniform(0.4, 0.6)),
    }
    shape = (int(rng.integers(1, 9)), int(rng.integers(1, 40)), int(rng.integers(1, 3)), int(rng.integers(1, 9)))
    raw = PC.kwta_inputs(8000 + seed, 2, shape) * np.float32(rng.uniform(0.2, 3.0))
    with backend.emulated("plain"):
        k, ko = PC._kwta_pair(orc, **over)
        for pool in (True, False):
            st = (rng.uniform(0, 0.3, (2, shape[0] * shape[1], 2)).astype(np.float32) if pool and seed % 2 else None)
            st_o = st.copy() if st is not None else None
            act, cyc = kwta.kwta_batch_host(k, raw, pool=pool, state=st)
            for i in range(2):
                ref, c = (orc.kwta_pool(ko, raw[i], st_o[i] if st_o is not None else None) if pool
                          else orc.kwta_layer(ko, raw[i]))
                assert np.array_equal(act[i], ref, equal_nan=True), (seed, over, shape, pool, i)
                assert cyc[i] == c
            if st is not None:
                assert np.array_equal(st, st_o)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_gabor_geometry_fuzz(orc, cdt):
    """16 seeded random Convolve geometries through both gabor kernels (the GPU tier runs 64)"""
    import backend
    import parity_cases as PC
    with backend.emulated("plain"):
        for seed in range(16):
            PC.case_gabor_fuzz(orc, seed, cdt)
