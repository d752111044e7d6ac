#!/usr/bin/env python3
"""Run a script of this repo against an experimental build of the library (auditory_amd/libauditory_hip_<tag>.so, built by
`python -m auditory_amd.build --tag <tag> -D...`):   python tools/run_with_lib.py <tag> bench.py --workload cfg1 ...
Tuning / A-B runs only; the shipped library is what every other entry point loads."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from auditory_amd import capi  # noqa: E402

tag, script = sys.argv[1], sys.argv[2]
capi.LIB_PATH = capi.LIB_PATH.replace(".so", "_%s.so" % tag)
assert os.path.exists(capi.LIB_PATH), capi.LIB_PATH
sys.argv = [script] + sys.argv[3:]
runpy.run_path(os.path.join(ROOT, script), run_name="__main__")
