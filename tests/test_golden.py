"""tests/golden/*.npz: regression vectors written by tests/golden/make_golden.py (see its docstring
for what they are and are not).  CPU tier: the oracle reproduces them bit for bit and the
emulator build of the kernels matches them; GPU tier: the shipped library matches them."""
import os
import sys

import pytest

import golden_cases as GC
from auditory_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emul"))


@pytest.mark.parametrize("name", GC.NAMES)
def test_oracle_reproduces_golden(orc, name):
    GC.check_oracle_reproduces(name)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
@pytest.mark.parametrize("name", GC.NAMES)
def test_emulated_kernels_match_golden(orc, name, cdt):
    import backend
    with backend.emulated("plain"):
        GC.check_library_against_golden(name, cdt)


@pytest.mark.gpu
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
@pytest.mark.parametrize("name", GC.NAMES)
def test_gpu_matches_golden(orc, name, cdt):
    import torch
    assert torch.cuda.is_available()
    GC.check_library_against_golden(name, cdt)


@pytest.mark.parametrize("name", GC.NAMES)
def test_reference_dumps_pin_the_oracle(orc, name):
    """Only where go/cmd/refdump has been run (a machine with Go and the reference's module cache): the REAL reference's
    outputs against the oracle, and against the kernels (emulator build).  Skipped -- parity stays "unpinned" -- otherwise."""
    if not GC.reference_dumps(name):
        pytest.skip("no reference dumps under tests/golden/ref (go/cmd/refdump has not been run on this machine)")
    assert GC.check_oracle_against_reference(name) > 0
    if GC.kwta_params_file(name):
        assert GC.check_kwta_defaults_against_reference(name) > 0     # aud_kwta_defaults and the oracle's, field by field
    import backend
    with backend.emulated("plain"):
        assert GC.check_library_against_reference(name, capi.AUD_F64) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", GC.NAMES)
def test_gpu_matches_reference_dumps(orc, name):
    if not GC.reference_dumps(name):
        pytest.skip("no reference dumps under tests/golden/ref (go/cmd/refdump has not been run)")
    for cdt in (capi.AUD_F64, capi.AUD_F32):
        assert GC.check_library_against_reference(name, cdt) > 0


def test_reference_hook_reads_what_refdump_writes(orc, tmp_path, monkeypatch):
    """The reader side of the pin, exercised with SIMULATED dumps (the oracle's own outputs written in refdump's file
    format: little-endian float64 / float32, row-major; the k-WTA parameter block as Go's %+v prints a struct): proves the
    hook finds, reshapes and compares every kind refdump writes -- mel, log-power, Energy, MFCC, deltas, delta-deltas, raw
    gabor, GborKwta at pool and layer level, kwta.KWTA.Defaults() -- it is NOT a pin (that needs the real reference's files)."""
    import numpy as np
    import make_golden as G
    from oracle import oracle as orc_mod
    import workloads as W
    name = "sndenv_16k_n400_nf32"
    oc, sig, pcm, items, gab = G.inputs(name)
    k = orc_mod.gabor_to_tensor(W.DEFAULT_GABOR_SPECS, 9, 9)
    kw = orc_mod.kwta_defaults()
    state = {}
    for r, s in items[:4]:
        o = orc_mod.process_segment_mfcc(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig[r], segment=s)
        pre = "%s_r%d_s%d_" % (name, r, s)
        for key, ext in (("mel_seg", "mel.f64"), ("log_power_seg", "logpower.f64"), ("energy", "energy.f64"), ("mfcc", "mfcc.f64"),
                         ("deltas", "mfccdeltas.f64"), ("delta_deltas", "mfccdeltadeltas.f64")):
            o[key].astype("<f8").tofile(str(tmp_path / (pre + ext)))
        g = np.zeros(G.GABOR[gab] + (2, 8), np.float32)
        assert orc_mod.gabor_convolve(o["mel_seg"], k, 3, 3, 2.0, g) == 0
        g.astype("<f4").tofile(str(tmp_path / (pre + "gabor.f32")))
        st = state.setdefault(r, np.zeros((16, 2), np.float32))       # carried across the job's segments, like se.Inhibs
        orc_mod.kwta_pool(kw, g, st)[0].astype("<f4").tofile(str(tmp_path / (pre + "kwtapool.f32")))
        orc_mod.kwta_layer(kw, g)[0].astype("<f4").tofile(str(tmp_path / (pre + "kwtalayer.f32")))
    (tmp_path / ("%s_r0_kwta_params.txt" % name)).write_text(GC.go_print_kwta(kw, orc_mod.kwta_update(kw)))
    monkeypatch.setattr(GC, "REF_DIR", str(tmp_path))
    dumps = GC.reference_dumps(name)
    assert len(dumps) == 4 and all(len(f) == len(GC.REF_KINDS) for _, _, f in dumps)
    assert GC.check_oracle_against_reference(name) == 4
    parsed = GC.parse_go_struct(open(GC.kwta_params_file(name)).read())
    assert parsed["On"] == "true" and "LayFFFB.Gi" in parsed and "Erev.K" in parsed and "XX1.InterpVal" in parsed
    assert GC.check_kwta_defaults_against_reference(name) >= 40        # 38 settable fields + the derived ones Go prints
    import backend
    with backend.emulated("plain"):
        assert GC.check_library_against_reference(name, capi.AUD_F64) == 4
    # ... and a dump that disagrees is caught, kind by kind
    bad = np.fromfile(dumps[1][2]["kwtapool"], "<f4")
    bad[np.argmax(bad)] += 1e-3
    bad.tofile(dumps[1][2]["kwtapool"])
    with pytest.raises(AssertionError, match="KWTAPool"):
        GC.check_oracle_against_reference(name)
    txt = open(GC.kwta_params_file(name)).read().replace("Iters:", "Iters:1", 1)
    open(GC.kwta_params_file(name), "w").write(txt)
    with pytest.raises(AssertionError, match="Iters"):
        GC.check_kwta_defaults_against_reference(name)
