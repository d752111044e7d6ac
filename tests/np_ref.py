"""Independent numpy float64 model of the hot path (uses numpy.fft, vectorised).

Written separately from oracle/auditory_oracle.c so that the two can be checked
against each other (tests/test_oracle.py).  Follows the same reference lines:
dft/dft.go:62-85, mel/mel.go:77-153, agabor/gabor.go:225-315, sound/sndenv.go:438-478.
"""
import numpy as np


def go_round(x):
    return int(np.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))


def msec_to_samples(ms, rate):
    return go_round(ms * 0.001 * rate)


def mel_table(nf, lo, hi, n, sr):
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    inv = lambda m: 700.0 * (np.exp(m / 1127.0) - 1.0)
    pts = mel(lo) + np.arange(nf + 2) * ((mel(hi) - mel(lo)) / (nf + 1))
    hz = inv(pts)
    bins = np.floor((n + 1) * hz / sr).astype(np.int32)
    return bins, hz


def mel_weights_dense(bins, nf, H):
    """dense [nf, H] weights for NON-degenerate, non-spilling tables"""
    W = np.zeros((nf, H))
    for f in range(nf):
        lo, c, hi = int(bins[f]), int(bins[f + 1]), int(bins[f + 2])
        with np.errstate(invalid="ignore", divide="ignore"):
            for b in range(lo, c + 1):
                W[f, b] = np.float64(b - lo) / np.float64(c - lo)
            for b in range(c + 1, hi + 1):
                W[f, b] = np.float64(hi - b) / np.float64(hi - c)
    return W


def frames(signal, n, step, T, border, seg_start=0):
    """[T, n] frame matrix + valid mask, sndenv.go:455-478 semantics"""
    L = len(signal)
    out = np.zeros((T, n))
    valid = np.zeros(T, bool)
    for s in range(T):
        st = seg_start + step * (s - border)
        en = st + n
        if en > L:
            break
        valid[s] = True
        if en <= 0:
            continue
        a = max(st, 0)
        out[s, a - st:] = signal[a:en]
    return out, valid


def melspec(signal, n, step, T, border, nf, lo, hi, sr, log_off=0.0, log_min=-10.0,
            dft_log_off=1.0, seg_start=0):
    H = n // 2 + 1
    fr, valid = frames(np.asarray(signal, np.float64), n, step, T, border, seg_start)
    X = np.fft.fft(fr, axis=1)[:, :H]
    P = X.real ** 2 + X.imag ** 2
    bins, _ = mel_table(nf, lo, hi, n, sr)
    W = mel_weights_dense(bins, nf, H)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = P @ W.T + log_off
        m = np.where(s == 0, log_min, np.log(s))
        lp = np.log(P + dft_log_off)
    m[~valid] = 0
    P[~valid] = 0
    lp[~valid] = 0
    return m.T.copy(), P.T.copy(), lp.T.copy()   # [nf,T], [H,T], [H,T]


def gabor4(mel, k, stx, sty, gain, py, px):
    nf, T = mel.shape
    ng, sy, sx = k.shape
    tmax = min(px * stx, T - stx)
    fmax = min(py * sty, nf - sty)
    out = np.zeros((py, px, 2, ng), np.float32)
    m = np.where(np.isnan(mel), 0.5, mel)
    for ti, t in enumerate(range(0, tmax, stx)):
        for fi, f in enumerate(range(0, fmax, sty)):
            patch = m[f:f + sy, t:t + sx]
            for g in range(ng):
                s = 0.0
                for a in range(sy):
                    for b in range(sx):
                        s += k[g, a, b] * patch[a, b]
                act = gain * abs(s)
                if s >= 0:
                    out[fi, ti, 0, g] = act
                else:
                    out[fi, ti, 1, g] = act
    return out
