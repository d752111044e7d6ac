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
    format: little-endian float64 / float32, row-major): proves the hook finds, reshapes and compares them -- it is NOT a
    pin (that needs the real reference's files)."""
    import numpy as np
    import make_golden as G
    from oracle import oracle as orc_mod
    import workloads as W
    name = "sndenv_16k_n400_nf32"
    oc, sig, pcm, items, gab = G.inputs(name)
    k = orc_mod.gabor_to_tensor(W.DEFAULT_GABOR_SPECS, 9, 9)
    for r, s in items[:2]:
        o = orc_mod.process_segment(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig[r], segment=s)
        o["mel_seg"].astype("<f8").tofile(str(tmp_path / ("%s_r%d_s%d_mel.f64" % (name, r, s))))
        o["log_power_seg"].astype("<f8").tofile(str(tmp_path / ("%s_r%d_s%d_logpower.f64" % (name, r, s))))
        g = np.zeros(G.GABOR[gab] + (2, 8), np.float32)
        assert orc_mod.gabor_convolve(o["mel_seg"], k, 3, 3, 2.0, g) == 0
        g.astype("<f4").tofile(str(tmp_path / ("%s_r%d_s%d_gabor.f32" % (name, r, s))))
    monkeypatch.setattr(GC, "REF_DIR", str(tmp_path))
    assert len(GC.reference_dumps(name)) == 2
    assert GC.check_oracle_against_reference(name) == 2
    import backend
    with backend.emulated("plain"):
        assert GC.check_library_against_reference(name, capi.AUD_F64) == 2
