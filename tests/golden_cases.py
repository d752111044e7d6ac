"""Golden-fixture checks shared by the CPU (oracle, emulator) and GPU test files."""
import os
import sys

import numpy as np

import workloads as W
from auditory_amd import capi, runtime

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G  # noqa: E402

NAMES = list(G.FIXTURES)


def load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def check_oracle_reproduces(name):
    """oracle today == oracle when the fixture was written (bit for bit)"""
    gold, now = load(name), G.compute(name)
    for k in gold.files:
        assert np.array_equal(gold[k], now[k], equal_nan=True), (name, k)


def check_library_against_golden(name, compute_dtype):
    """the shipped C ABI (GPU, or the emulator build in CPU tests) against the stored outputs"""
    gold = load(name)
    oc, sig, pcm, items, gab = G.inputs(name)
    assert int(pcm.astype(np.int64).sum()) == int(gold["pcm_crc"][0]), "seeded input drifted"
    L = sig.shape[1]
    its = runtime.make_items([r * L for r, s in items], [L] * len(items),
                             [s * oc.sp.stride_samples for r, s in items])
    gspec = dict(size=(9, 9), stride=(3, 3), gain=2.0, specs=W.DEFAULT_GABOR_SPECS) if gab else None
    plan = W.product_plan(oc, compute_dtype, gspec)
    try:
        mel, _, lp = plan.melspec_host(sig.ravel(), its, False, True)
        ok, msg = W.feature_close(mel, gold["mel"], compute_dtype, lin_axis=1)
        assert ok, (name, "mel", msg)
        if compute_dtype == capi.AUD_F64:
            ok, msg = W.close_enough(lp, gold["log_power"], 3e-7)
        else:  # f32 FFT: bins are accurate relative to the frame's peak bin (see spectrum_close)
            ok, msg = W.spectrum_close(lp, gold["log_power"], 4e-6, log_offset=1.0)
        assert ok, (name, "log_power", msg)
        if gab:
            py, px = G.GABOR[gab]
            out = np.zeros((len(items), py, px, 2, 8), np.float32)
            plan.gabor_host(mel, out)            # gabor of the library's own mel, as SndEnv does
            ok, msg = W.feature_close(out, gold["gabor"], compute_dtype, tol_f64=W.TOL_F64_GABOR)
            assert ok, (name, "gabor", msg)
        if "mfcc" in gold.files and compute_dtype == capi.AUD_F64:
            # the speech fixtures carry the rest of ProcessSegment (sndenv.go:360-432): one call of the MFCC entry point
            tplan = W.product_plan(oc, compute_dtype, mfcc_coefs=13)
            try:
                o = tplan.melspec_mfcc_host(sig.ravel(), its)
            finally:
                tplan.close()
            fused = tplan.kernel_name in ("w20x10", "w16x16")
            tols = dict(mel=1e-5, energy=1e-5, mfcc=1e-5, deltas=1e-5 if fused else 2e-5, delta_deltas=5e-5 if fused else 2e-4)
            for key, tol in tols.items():
                ok, msg = W.close_enough(o[key], gold[key], tol)
                assert ok, (name, key, msg)
        if gab and compute_dtype == capi.AUD_F32:
            # k-WTA of the STORED gabor tensor: float32 in the reference's order whatever the plan computes in, so
            # exact (checked once, with the float32 plans)
            from auditory_amd import kwta
            k = kwta.KWTA()
            k.Defaults()
            act, _ = kwta.kwta_batch_host(k, gold["gabor"], pool=True)
            assert np.array_equal(act, gold["kwta"]), (name, "kwta")
    finally:
        plan.close()


REF_DIR = os.path.join(HERE, "golden", "ref")


def reference_dumps(name):
    """[(row, segment, {kind: path})] of what go/cmd/refdump wrote for fixture `name` (empty where it has not been run)"""
    cfg, seg_ms, dur, rows, segs, seed, gab = G.FIXTURES[name]
    found = []
    for r in range(rows):
        for s in segs:
            files = {k: os.path.join(REF_DIR, "%s_r%d_s%d_%s" % (name, r, s, ext))
                     for k, ext in (("mel", "mel.f64"), ("logpower", "logpower.f64"), ("gabor", "gabor.f32"))}
            files = {k: p for k, p in files.items() if os.path.exists(p)}
            if "mel" in files:
                found.append((r, s, files))
    return found


def check_oracle_against_reference(name):
    """THE pin: outputs of the real reference (emer/auditory, dumped by go/cmd/refdump from the WAVs of
    tests/golden/make_ref_inputs.py) against the oracle on the same samples -- two float64 implementations of the same
    arithmetic with different FFTs: 1e-9 of max(1, |ref|)"""
    from oracle import oracle as orc
    oc, sig, pcm, items, gab = G.inputs(name)
    n = 0
    for r, s, files in reference_dumps(name):
        o = orc.process_segment(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig[r], segment=s)
        ref_mel = np.fromfile(files["mel"], "<f8").reshape(oc.nf, oc.T)
        ok, msg = W.close_enough(o["mel_seg"], ref_mel, 1e-9)
        assert ok, (name, r, s, "mel", msg)
        if "logpower" in files:
            ref_lp = np.fromfile(files["logpower"], "<f8").reshape(oc.H, oc.T)
            ok, msg = W.close_enough(o["log_power_seg"], ref_lp, 1e-9)
            assert ok, (name, r, s, "log_power", msg)
        if gab and "gabor" in files:
            py, px = G.GABOR[gab]
            ref_g = np.fromfile(files["gabor"], "<f4").reshape(py, px, 2, 8)
            g = np.zeros((py, px, 2, 8), np.float32)
            assert orc.gabor_convolve(o["mel_seg"], orc.gabor_to_tensor(W.DEFAULT_GABOR_SPECS, 9, 9), 3, 3, 2.0, g) == 0
            ok, msg = W.close_enough(g, ref_g, 1e-6)
            assert ok, (name, r, s, "gabor", msg)
        n += 1
    return n


def check_library_against_reference(name, compute_dtype):
    """the shipped C ABI (GPU, or the emulator build) against the real reference's dumps, north-star criterion"""
    oc, sig, pcm, items, gab = G.inputs(name)
    dumps = reference_dumps(name)
    if not dumps:
        return 0
    L = sig.shape[1]
    its = runtime.make_items([r * L for r, s, _ in dumps], [L] * len(dumps), [s * oc.sp.stride_samples for r, s, _ in dumps])
    plan = W.product_plan(oc, compute_dtype)
    try:
        mel, _, _ = plan.melspec_host(sig.ravel(), its)
    finally:
        plan.close()
    ref = np.stack([np.fromfile(f["mel"], "<f8").reshape(oc.nf, oc.T) for _, _, f in dumps])
    if compute_dtype == capi.AUD_F64:
        ok, msg = W.close_enough(mel, ref, 1e-5)
    else:
        ok, msg = W.feature_close(mel, ref, compute_dtype, lin_axis=1)
    assert ok, (name, "mel vs the reference", msg)
    return len(dumps)
