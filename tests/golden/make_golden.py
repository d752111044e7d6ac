"""Generates tests/golden/*.npz from the CPU oracle (oracle/auditory_oracle.c, kwta_oracle.c).

The reference (Go) cannot run in this pipeline and ships no vectors of its own, so these
fixtures are REGRESSION vectors of the oracle, not reference outputs: they freeze today's oracle
(itself pinned by the KATs and the numpy cross-check of tests/test_oracle.py) so that later edits
to oracle/ or to the kernels cannot drift silently.  Inputs are regenerated from the seeds;
expected outputs are stored.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import workloads as W  # noqa: E402
from auditory_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

# name -> (config, segment_ms override, seconds of audio, rows, segments, seed id, gabor?)
FIXTURES = {
    "sndenv_16k_n400_nf32": ("sndenv_16k_n400_nf32", None, 0.45, 2, [0, 1, 3], 21, "pool8x2"),
    "cfg2_16k_n400_nf40": ("cfg2_16k_n400_nf40", None, 1.0, 2, [0], 22, "pool11x32"),
    "cfg2_16k_n512_nf40": ("cfg2_16k_n512_nf40", None, 1.0, 2, [0], 23, "pool11x32"),
    "cfg1_44k_n1103_nf32": ("cfg1_44k_n1103_nf32", None, 0.3, 1, [0, 1], 24, "pool8x2"),
    "cfg5_44k_n2048_nf128": ("cfg5_44k_n2048_nf128", 300.0, 0.4, 2, [0], 25, None),
    # SURVEY 8d's cfg-1 input (configs[0]: examples/processspeech on one speech WAV, processspeech.go:190-283): 3 s of
    # synth.speech_like -- shaped noise / pulse trains under a 4 Hz envelope with exact-zero gaps -- at 16 kHz (N = 400) and at
    # the shipped WAVs' 44.1 kHz (N = 1103), SndEnv defaults, gabor through the 4-D shape [8,2,2,8] (Q9), with the MFCC tail.
    # Segments: 0 (left pad + leading gap), 1 (first syllable), 2 (a gap: all-zero frames), 5 (voiced), 12, 29 (the last).
    "speech_16k_n400_nf32": ("sndenv_16k_n400_nf32", None, 3.0, 1, [0, 1, 2, 5, 12, 29], 31, "pool8x2"),
    "speech_44k_n1103_nf32": ("cfg1_44k_n1103_nf32", None, 3.0, 1, [0, 1, 2, 5, 12, 29], 32, "pool8x2"),
    # round 6: processspeech's parameters at other sample rates -- the any-N kernel's in-place route (8 kHz: eight frames per
    # workgroup; 48 kHz: two), the chirp kernel beyond N = 1103 (22.05 kHz: N = 551), radix 7 on an odd window (50 ms at 44.1 kHz)
    "rate_8k_n200_nf32": ("rate_8k_n200_nf32", None, 0.3, 2, [0, 1, 2], 41, "pool8x2"),
    "rate_22k_n551_nf32": ("rate_22k_n551_nf32", None, 0.3, 2, [0, 1], 42, "pool8x2"),
    "rate_48k_n1200_nf32": ("rate_48k_n1200_nf32", None, 0.3, 2, [0, 1], 43, "pool8x2"),
    "win50_44k_n2205_nf64": ("win50_44k_n2205_nf64", None, 0.3, 1, [0, 1], 44, None),
}
GABOR = {"pool8x2": (8, 2), "pool11x32": (11, 32)}
SPEECH = ("speech_16k_n400_nf32", "speech_44k_n1103_nf32")   # inputs from synth.speech_like; fixtures carry the MFCC tail


def inputs(name):
    cfg, seg_ms, dur, rows, segs, seed, gab = FIXTURES[name]
    oc = W.OracleCfg(orc, cfg, seg_ms)
    L = int(dur * oc.sr)
    if name in SPEECH:
        sig, pcm = synth.speech_like(seed, L, oc.sr)
        sig, pcm = sig[None], pcm[None]
    else:
        sig, pcm = synth.batch(seed, rows, L, oc.sr)
    return oc, sig, pcm, [(r, s) for r in range(rows) for s in segs], gab


def compute(name):
    oc, sig, pcm, items, gab = inputs(name)
    mel, lp = [], []
    for r, s in items:
        o = orc.process_segment(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig[r], segment=s)
        mel.append(o["mel_seg"])
        lp.append(o["log_power_seg"])
    out = dict(mel=np.stack(mel), log_power=np.stack(lp).astype(np.float32),
               bin_pts=oc.bins, pcm_crc=np.array([int(pcm.astype(np.int64).sum())]))
    if name in SPEECH:   # the rest of ProcessSegment (sndenv.go:360-432): Energy, MFCC with row 0 <- Energy, deltas, delta-deltas
        tail = [orc.process_segment_mfcc(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig[r], segment=s) for r, s in items]
        for k in ("mfcc", "deltas", "delta_deltas", "energy"):
            out[k] = np.stack([t[k] for t in tail])
        assert all(np.array_equal(t["mel_seg"], m) for t, m in zip(tail, mel))
    if gab:
        py, px = GABOR[gab]
        k = orc.gabor_to_tensor(W.DEFAULT_GABOR_SPECS, 9, 9)
        g = np.zeros((len(items), py, px, 2, 8), np.float32)
        for i in range(len(items)):
            assert orc.gabor_convolve(out["mel"][i], k, 3, 3, 2.0, g[i]) == 0
        out["gabor"] = g
        out["gabor_k"] = k
        # SndEnv.ApplyKwta with the default parameters (KWTAPool, fresh Inhibs per item): float32, bit-exact
        kw = orc.kwta_defaults()
        out["kwta"] = np.stack([orc.kwta_pool(kw, g[i])[0] for i in range(len(items))])
    return out


if __name__ == "__main__":
    for name in (sys.argv[1:] or FIXTURES):   # (names on the command line: only those)
        d = compute(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, {k: v.shape for k, v in d.items()})
