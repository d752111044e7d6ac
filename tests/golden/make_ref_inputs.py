"""Writes the inputs of go/cmd/refdump (the program that can pin the oracle against the REAL reference where Go and the
module cache exist): one 16-bit mono WAV per row of every golden fixture (the same seeded PCM the fixtures use) under
tests/golden/ref_in/, and ref_in/jobs.txt with one refdump job per WAV.  refdump's outputs go to tests/golden/ref/; when
they exist, tests/test_golden.py checks the oracle and the HIP path against them.  Run: python tests/golden/make_ref_inputs.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import make_golden as G  # noqa: E402
import workloads as W  # noqa: E402
from auditory_amd import sound  # noqa: E402


def main():
    out = os.path.join(HERE, "ref_in")
    os.makedirs(out, exist_ok=True)
    os.makedirs(os.path.join(HERE, "ref"), exist_ok=True)
    jobs = []
    for name, (cfg, seg_ms, dur, rows, segs, seed, gab) in G.FIXTURES.items():
        oc, sig, pcm, items, _ = G.inputs(name)
        sr, win, step, seg, stride, border, nf, lo, hi = W.CONFIGS[cfg]
        if seg_ms is not None:
            seg = stride = seg_ms
        py, px = G.GABOR[gab] if gab else (0, 0)
        for r in range(rows):
            w = sound.Wave()
            w.Data, w.SourceBitDepth, w._rate, w._channels = pcm[r].astype("int64"), 16, sr, 1
            fn = os.path.join(out, "%s_r%d.wav" % (name, r))
            assert w.WriteWave(fn) is None
            jobs.append("%s %s %g %g %g %g %d %d %g %g %d %d %s" % (
                os.path.relpath(fn, ROOT), os.path.join("tests", "golden", "ref", "%s_r%d" % (name, r)), win, step, seg,
                stride, border, nf, lo, hi, py, px, ",".join(str(s) for s in segs)))
    with open(os.path.join(out, "jobs.txt"), "w") as fh:
        fh.write("\n".join(jobs) + "\n")
    print("wrote %d jobs; on a machine with Go: (cd %s && go run ./go/cmd/refdump @tests/golden/ref_in/jobs.txt)" % (len(jobs), ROOT))


if __name__ == "__main__":
    main()
