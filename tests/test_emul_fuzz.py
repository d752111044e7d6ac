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


@pytest.mark.parametrize("seed", range(48))
def test_random_configs_vs_oracle(orc, seed):
    import backend
    import parity_cases as PC
    with backend.emulated("plain"):
        PC.case_random_any_n(orc, seed, WINDOWS)


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
