"""Seeded synthetic inputs shared by tests and bench.py.

White Gaussian noise (sigma 0.1) + 3 random tones, clipped to [-1, 1], quantised
to int16 and divided by 0x7FFF exactly as sound/sound.go:130-141 does for 16-bit
PCM.  Generator: PCG64, seed = 20260000 + config_id*1000 + utterance_idx.
"""
import numpy as np


def utterance_pcm(config_id, idx, n_samples, sr):
    """int16 PCM of one synthetic mono utterance"""
    rng = np.random.Generator(np.random.PCG64(20260000 + config_id * 1000 + idx))
    x = rng.normal(0.0, 0.1, n_samples)
    t = np.arange(n_samples) / float(sr)
    hi = min(7000.0, 0.45 * sr)
    for _ in range(3):
        f = rng.uniform(100.0, hi)
        a = rng.uniform(0.05, 0.3)
        ph = rng.uniform(0, 2 * np.pi)
        x += a * np.sin(2 * np.pi * f * t + ph)
    x = np.clip(x, -1.0, 1.0)
    return np.round(x * 32767.0).astype(np.int16)


def pcm_to_float64(pcm):
    """sound/sound.go:138: float64(v) / float64(0x7FFF)"""
    return pcm.astype(np.float64) / float(0x7FFF)


def batch(config_id, n_utt, n_samples, sr, row_len=None, first_idx=0):
    """[n_utt, row_len] float64 batch (zero tail beyond n_samples) + int16 PCM"""
    row_len = row_len or n_samples
    pcm = np.zeros((n_utt, row_len), np.int16)
    for i in range(n_utt):
        pcm[i, :n_samples] = utterance_pcm(config_id, first_idx + i, n_samples, sr)
    return pcm_to_float64(pcm), pcm
