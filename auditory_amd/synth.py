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


def speech_like_pcm(seed, n_samples, sr):
    """SURVEY 8d's cfg-1 input: int16 PCM of a TIMIT-style mono utterance -- speech-band shaped noise under a 4 Hz syllable
    envelope (what configs[0], examples/processspeech on one speech WAV, sees: processspeech.go:190-283).  Built so that the
    dynamics the tone + noise-floor batches lack are there:
      * spectral shape: three formant-like resonances (500 / 1500 / 2500 Hz) over a -12 dB/octave tilt above 500 Hz and a
        low cut under 100 Hz -- >= 60 dB between the spectral peak and 7 kHz before quantisation (the int16 rounding then sets
        a white floor near -100 dB re full scale, as in a real recording);
      * 4 Hz amplitude modulation: syllables of 170 ms (raised-cosine edges, levels drawn per syllable over 20 dB) separated by
        80 ms of EXACT zeros -- whole frames of zero samples, for which dft.Power stores ln(0 + LogOffSet) and mel.FilterDft
        the LogMin of an exactly-zero sum; the utterance starts and ends inside a gap;
      * a voiced source in every other syllable: a 110-140 Hz pulse train through the same resonances (harmonic structure:
        narrow peaks the low mel triangles resolve) instead of noise.
    Generator PCG64(20260000 + seed); the samples are what a WAV of it holds (sound.go:130-141 divides by 0x7FFF)."""
    rng = np.random.Generator(np.random.PCG64(20260000 + int(seed)))
    n = int(n_samples)
    t = np.arange(n) / float(sr)
    f = np.fft.rfftfreq(n, 1.0 / sr)
    shape = np.zeros_like(f)
    for fc, bw, g in ((500.0, 120.0, 1.0), (1500.0, 180.0, 0.4), (2500.0, 250.0, 0.15)):
        shape += g / np.sqrt(1.0 + ((f - fc) / bw) ** 2)
    shape *= 1.0 / (1.0 + (f / 500.0) ** 2)                      # -12 dB / octave above 500 Hz
    shape *= (f / 100.0) ** 2 / (1.0 + (f / 100.0) ** 2)         # low cut
    noise = np.fft.irfft(np.fft.rfft(rng.normal(0.0, 1.0, n)) * shape, n)
    period, on, edge = 0.25, 0.17, 0.03
    x = np.zeros(n)
    k = 0
    start = 0.04                                                  # the first 40 ms are a gap
    while start + on < t[-1] - 0.02:
        lo, hi = int(round(start * sr)), int(round((start + on) * sr))
        tt = (np.arange(lo, hi) - lo) / float(sr)
        env = np.ones(hi - lo)
        ne = int(round(edge * sr))
        ramp = 0.5 - 0.5 * np.cos(np.pi * np.arange(ne) / ne)
        env[:ne], env[-ne:] = ramp, ramp[::-1]
        level = 10.0 ** (-rng.uniform(0.0, 20.0) / 20.0)
        if k % 2 == 1:                                            # voiced: glottal pulse train through the resonances
            f0 = rng.uniform(110.0, 140.0)
            src = np.zeros(hi - lo)
            src[(np.arange(0.0, on, 1.0 / f0) * sr).astype(int)] = 1.0
            seg = np.fft.irfft(np.fft.rfft(src, n) * shape, n)[:hi - lo]
        else:
            seg = noise[lo:hi]
        seg = seg / (np.abs(seg).max() + 1e-30)
        x[lo:hi] = 0.6 * level * env * seg
        start += period
        k += 1
    pcm = np.round(np.clip(x, -1.0, 1.0) * 32767.0).astype(np.int16)
    return pcm


def speech_like(seed, n_samples, sr):
    """(float64 signal as Wave.SoundToTensor gives it, int16 PCM) of speech_like_pcm"""
    pcm = speech_like_pcm(seed, n_samples, sr)
    return pcm_to_float64(pcm), pcm
