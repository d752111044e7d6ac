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
            wave = tplan.kernel_name in ("w20x10", "w16x16")      # (float32 spectrum behind the tail; the any-N kernel: float64)
            tols = dict(mel=1e-5, energy=1e-5, mfcc=1e-5, deltas=1e-5, delta_deltas=5e-5 if wave else 1e-5)
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
# what go/cmd/refdump writes per (row, segment): kind -> file suffix
REF_KINDS = (("mel", "mel.f64"), ("logpower", "logpower.f64"), ("energy", "energy.f64"), ("mfcc", "mfcc.f64"),
             ("deltas", "mfccdeltas.f64"), ("delta_deltas", "mfccdeltadeltas.f64"), ("gabor", "gabor.f32"),
             ("kwtapool", "kwtapool.f32"), ("kwtalayer", "kwtalayer.f32"))


def reference_dumps(name):
    """[(row, segment, {kind: path})] of what go/cmd/refdump wrote for fixture `name` (empty where it has not been run), in
    the job's segment order (the order KWTAPool's carried state was produced in)"""
    cfg, seg_ms, dur, rows, segs, seed, gab = G.FIXTURES[name]
    found = []
    for r in range(rows):
        for s in segs:
            files = {k: os.path.join(REF_DIR, "%s_r%d_s%d_%s" % (name, r, s, ext)) for k, ext in REF_KINDS}
            files = {k: p for k, p in files.items() if os.path.exists(p)}
            if "mel" in files:
                found.append((r, s, files))
    return found


def parse_go_struct(text):
    """Go's fmt %+v of a struct -- "{On:true Iters:20 LayFFFB:{On:true Gi:1.5 ...} Gbar:{E:0.5 ...} ...}" -- as a flat
    dict {"On": "true", "LayFFFB.Gi": "1.5", ...}"""
    out, stack, i, text = {}, [], 0, text.strip()
    while i < len(text):
        c = text[i]
        if c in "{ ":
            i += 1
        elif c == "}":
            if stack:
                stack.pop()
            i += 1
        else:
            j = text.index(":", i)
            key = text[i:j]
            if text[j + 1] == "{":
                stack.append(key)
                i = j + 2
            else:
                k = j + 1
                while k < len(text) and text[k] not in " }":
                    k += 1
                out[".".join(stack + [key])] = text[j + 1:k]
                i = k
    return out


# kwta.KWTA (emer/vision v1.1.15) / fffb.Params, nxx1.Params, chans.Chans (emer/leabra v1.1.48) field paths, as Go prints
# them, -> the field of aud_kwta_params / oracle.Kwta that restates it.  The derived fields (FBDt, SigGainNVar, SigMultEff,
# SigValAt0, InterpVal, ErevSubThr, ThrSubErev, ActDt) map to what the oracle's Update computes.
def _kwta_field_map():
    m = {"On": ("on",), "Iters": ("iters",), "DelActThr": ("del_act_thr",), "ActTau": ("act_tau",)}
    for go, c in (("LayFFFB", "lay"), ("PoolFFFB", "pool")):
        for gf, cf in (("On", "on"), ("Gi", "gi"), ("FF", "ff"), ("FB", "fb"), ("FBTau", "fb_tau"), ("MaxVsAvg", "max_vs_avg"),
                       ("FF0", "ff0")):
            m["%s.%s" % (go, gf)] = (c, cf)
    for gf, cf in (("Thr", "thr"), ("Gain", "gain"), ("NVar", "nvar"), ("VmActThr", "vm_act_thr"), ("SigMult", "sig_mult"),
                   ("SigMultPow", "sig_mult_pow"), ("SigGain", "sig_gain"), ("InterpRange", "interp_range"),
                   ("GainCorRange", "gain_cor_range"), ("GainCor", "gain_cor")):
        m["XX1." + gf] = ("xx1", cf)
    for go, c in (("Gbar", "gbar"), ("Erev", "erev")):
        for i, ch in enumerate("ELIK"):
            m["%s.%s" % (go, ch)] = (c, i)
    return m


_KWTA_DERIVED = {"LayFFFB.FBDt": ("lay_fb_dt",), "PoolFFFB.FBDt": ("pool_fb_dt",), "XX1.SigGainNVar": ("sig_gain_nvar",),
                 "XX1.SigMultEff": ("sig_mult_eff",), "XX1.SigValAt0": ("sig_val_at0",), "XX1.InterpVal": ("interp_val",),
                 "ActDt": ("act_dt",)}
_KWTA_DERIVED.update({"ErevSubThr.%s" % ch: ("erev_sub_thr", i) for i, ch in enumerate("ELIK")})
_KWTA_DERIVED.update({"ThrSubErev.%s" % ch: ("thr_sub_erev", i) for i, ch in enumerate("ELIK")})


def _get(obj, path):
    for p in path:
        obj = obj[p] if isinstance(p, int) else getattr(obj, p)
    return obj


def kwta_params_file(name, row=0):
    p = os.path.join(REF_DIR, "%s_r%d_kwta_params.txt" % (name, row))
    return p if os.path.exists(p) else None


def check_kwta_defaults_against_reference(name):
    """the parameter block kwta.KWTA.Defaults() produced in the real reference (printed %+v by refdump) against
    oracle/kwta_oracle.c's orc_kwta_defaults AND the product's aud_kwta_defaults, field by field, as float32; the derived
    fields against the oracle's Update.  Returns the number of fields compared (0: no dump)."""
    from oracle import oracle as orc
    fn = kwta_params_file(name)
    if fn is None:
        return 0
    go = parse_go_struct(open(fn).read())
    ok_, prod = orc.kwta_defaults(), capi.KwtaParams()
    import ctypes
    capi.load().aud_kwta_defaults(ctypes.byref(prod))
    prod_names = {"lay": "lay_fffb", "pool": "pool_fffb"}
    n, fmap = 0, _kwta_field_map()
    missing = [k for k in fmap if k not in go]
    assert not missing, "%s: fields the restatement assumes that kwta.KWTA does not print: %s" % (fn, missing)
    for key, path in fmap.items():
        txt = go[key]
        want = {"true": 1.0, "false": 0.0}.get(txt)
        want = np.float32(float(txt)) if want is None else np.float32(want)
        got_o = np.float32(_get(ok_, path))
        got_p = np.float32(_get(prod, tuple(prod_names.get(p, p) if isinstance(p, str) else p for p in path)))
        assert got_o == want, "oracle kwta default %s = %r, the reference has %s" % (key, got_o, txt)
        assert got_p == want, "aud_kwta_defaults %s = %r, the reference has %s" % (key, got_p, txt)
        n += 1
    der = orc.kwta_update(ok_)
    for key, path in _KWTA_DERIVED.items():
        if key in go:   # (derived fields are compared where Go prints them)
            want, got = np.float32(float(go[key])), np.float32(_get(der, path))
            assert abs(float(got) - float(want)) <= 1e-6 * max(1.0, abs(float(want))), "derived %s: %r vs %s" % (key, got, go[key])
            n += 1
    unknown = sorted(set(go) - set(fmap) - set(_KWTA_DERIVED))
    if unknown:
        print("kwta.KWTA prints fields the restatement does not carry: %s" % unknown)
    return n


def check_oracle_against_reference(name):
    """THE pin: outputs of the real reference (emer/auditory, dumped by go/cmd/refdump from the WAVs of
    tests/golden/make_ref_inputs.py) against the oracle on the same samples -- two float64 implementations of the same
    arithmetic with different FFTs: 1e-9 of max(1, |ref|) for mel / log-power / Energy / MFCC / deltas / delta-deltas
    (f-1), 1e-6 for the float32 gabor tensor; the k-WTA stage (f-4: kwta_oracle.c was written from memory of emer/vision) on
    the REFERENCE'S raw gabor tensor, pool level with the state carried in the job's order and layer level: the same float32
    operations in the same order, so 2e-6 absolute is asked and bit-equality reported."""
    from oracle import oracle as orc
    oc, sig, pcm, items, gab = G.inputs(name)
    n = 0
    kw = orc.kwta_defaults()
    state, state_row = None, None
    exact = []
    for r, s, files in reference_dumps(name):
        o = orc.process_segment_mfcc(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig[r], segment=s)
        ref_mel = np.fromfile(files["mel"], "<f8").reshape(oc.nf, oc.T)
        ok, msg = W.close_enough(o["mel_seg"], ref_mel, 1e-9)
        assert ok, (name, r, s, "mel", msg)
        for kind, key, shape in (("logpower", "log_power_seg", (oc.H, oc.T)), ("energy", "energy", (oc.T,)),
                                 ("mfcc", "mfcc", (13, oc.T)), ("deltas", "deltas", (13, oc.T)),
                                 ("delta_deltas", "delta_deltas", (13, oc.T))):
            if kind in files:
                ref = np.fromfile(files[kind], "<f8").reshape(shape)
                ok, msg = W.close_enough(o[key], ref, 1e-9)
                assert ok, (name, r, s, kind, msg)
        if gab and "gabor" in files:
            py, px = G.GABOR[gab]
            ref_g = np.fromfile(files["gabor"], "<f4").reshape(py, px, 2, 8)
            g = np.zeros((py, px, 2, 8), np.float32)
            assert orc.gabor_convolve(o["mel_seg"], orc.gabor_to_tensor(W.DEFAULT_GABOR_SPECS, 9, 9), 3, 3, 2.0, g) == 0
            ok, msg = W.close_enough(g, ref_g, 1e-6)
            assert ok, (name, r, s, "gabor", msg)
            if "kwtapool" in files:
                if state_row != r:                       # a fresh SndEnv per job: se.Inhibs starts empty
                    state, state_row = np.zeros((py * px, 2), np.float32), r
                got, _ = orc.kwta_pool(kw, ref_g, state)
                ref_k = np.fromfile(files["kwtapool"], "<f4").reshape(ref_g.shape)
                assert np.abs(got - ref_k).max() <= 2e-6, (name, r, s, "KWTAPool", float(np.abs(got - ref_k).max()))
                exact.append(bool(np.array_equal(got, ref_k)))
            if "kwtalayer" in files:
                got, _ = orc.kwta_layer(kw, ref_g)
                ref_k = np.fromfile(files["kwtalayer"], "<f4").reshape(ref_g.shape)
                assert np.abs(got - ref_k).max() <= 2e-6, (name, r, s, "KWTALayer", float(np.abs(got - ref_k).max()))
                exact.append(bool(np.array_equal(got, ref_k)))
        n += 1
    if exact:
        print("%s: k-WTA oracle vs the reference: %d of %d tensors bit-identical" % (name, sum(exact), len(exact)))
    return n


def check_library_against_reference(name, compute_dtype):
    """the shipped C ABI (GPU, or the emulator build) against the real reference's dumps: mel under the north-star criterion;
    where the dumps hold them, the MFCC tail (float64 plans: the tolerances of case_speech_like_sndenv) and the k-WTA stage
    run by the LIBRARY on the reference's raw gabor tensor (float32 in the reference's order: 2e-6 absolute)"""
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
    if compute_dtype == capi.AUD_F64 and all("delta_deltas" in f for _, _, f in dumps):
        tplan = W.product_plan(oc, compute_dtype, mfcc_coefs=13)
        try:
            o = tplan.melspec_mfcc_host(sig.ravel(), its)
        finally:
            tplan.close()
        wave = tplan.kernel_name in ("w20x10", "w16x16")
        for kind, key, shape, tol in (("energy", "energy", (oc.T,), 1e-5), ("mfcc", "mfcc", (13, oc.T), 1e-5),
                                      ("deltas", "deltas", (13, oc.T), 1e-5),
                                      ("delta_deltas", "delta_deltas", (13, oc.T), 5e-5 if wave else 1e-5)):
            ref = np.stack([np.fromfile(f[kind], "<f8").reshape(shape) for _, _, f in dumps])
            ok, msg = W.close_enough(o[key], ref, tol)
            assert ok, (name, kind + " vs the reference", msg)
    if gab and all("kwtapool" in f and "gabor" in f for _, _, f in dumps):
        from auditory_amd import kwta
        k = kwta.KWTA()
        k.Defaults()
        py, px = G.GABOR[gab]
        state, state_row = None, None
        for r, s, f in dumps:
            raw = np.fromfile(f["gabor"], "<f4").reshape(py, px, 2, 8)
            if state_row != r:
                state, state_row = np.zeros((py * px, 2), np.float32), r
            act = raw.copy()
            k.KWTAPool(raw, act, state)
            ref_k = np.fromfile(f["kwtapool"], "<f4").reshape(raw.shape)
            assert np.abs(act - ref_k).max() <= 2e-6, (name, r, s, "library KWTAPool vs the reference")
            if "kwtalayer" in f:
                act = raw.copy()
                k.KWTALayer(raw, act)
                ref_k = np.fromfile(f["kwtalayer"], "<f4").reshape(raw.shape)
                assert np.abs(act - ref_k).max() <= 2e-6, (name, r, s, "library KWTALayer vs the reference")
    return len(dumps)


def go_print_kwta(kw, derived):
    """what Go's fmt %+v prints for a kwta.KWTA holding the oracle's parameter block (for the SIMULATED dump of the hook's
    self-test; field order and names as emer/vision v1.1.15 / emer/leabra v1.1.48 declare them, from memory)"""
    def v(x):
        return np.format_float_positional(np.float32(x), unique=True, trim="-")

    def fffb(p, dt):
        return "{On:%s Gi:%s FF:%s FB:%s FBTau:%s MaxVsAvg:%s FF0:%s FBDt:%s}" % (
            "true" if p.on else "false", v(p.gi), v(p.ff), v(p.fb), v(p.fb_tau), v(p.max_vs_avg), v(p.ff0), v(dt))

    def chans(c):
        return "{E:%s L:%s I:%s K:%s}" % tuple(v(c[i]) for i in range(4))
    x = kw.xx1
    xx1 = ("{Thr:%s Gain:%s NVar:%s VmActThr:%s SigMult:%s SigMultPow:%s SigGain:%s InterpRange:%s GainCorRange:%s GainCor:%s "
           "SigGainNVar:%s SigMultEff:%s SigValAt0:%s InterpVal:%s}") % tuple(v(a) for a in (
               x.thr, x.gain, x.nvar, x.vm_act_thr, x.sig_mult, x.sig_mult_pow, x.sig_gain, x.interp_range, x.gain_cor_range,
               x.gain_cor, derived.sig_gain_nvar, derived.sig_mult_eff, derived.sig_val_at0, derived.interp_val))
    return "{On:%s Iters:%d DelActThr:%s LayFFFB:%s PoolFFFB:%s XX1:%s ActTau:%s Gbar:%s Erev:%s ErevSubThr:%s ThrSubErev:%s ActDt:%s}\n" % (
        "true" if kw.on else "false", kw.iters, v(kw.del_act_thr), fffb(kw.lay, derived.lay_fb_dt), fffb(kw.pool, derived.pool_fb_dt),
        xx1, v(kw.act_tau), chans(kw.gbar), chans(kw.erev), chans(derived.erev_sub_thr), chans(derived.thr_sub_erev), v(derived.act_dt))
