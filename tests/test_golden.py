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
