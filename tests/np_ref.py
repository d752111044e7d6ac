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


# ---- k-WTA (float32 scalar restatement, independent of the C oracle's code; small tensors only) ----

F = np.float32


def fast_exp32(x):
    """goki/mat32 FastExp with numpy int32 arithmetic"""
    x = F(x)
    if x <= F(-88.76731):
        return F(0)
    i = np.int32(int(F(12102203.0) * x)) + np.int32(127 * (1 << 23))
    m = (int(i) >> 7) & 0xFFFF
    c = ((((((((((3537 * m) >> 16) + 13668) * m) >> 18) + 15817) * m) >> 14) - 80470) * m) >> 11
    return np.array([int(i) + c], np.int32).view(np.float32)[0]


class KwtaRef:
    def __init__(self, k):
        """k: any object with the oracle's Kwta field names"""
        self.k = k
        x = k.xx1
        self.sig_gain_nvar = F(x.sig_gain) / F(x.nvar)
        self.sig_mult_eff = F(x.sig_mult) * F(float(F(x.gain) * F(x.nvar)) ** float(F(x.sig_mult_pow)))
        self.sig_val_at0 = F(0.5) * self.sig_mult_eff
        self.interp_val = self.xx1_gain_cor(F(x.interp_range)) - self.sig_val_at0
        self.est = [F(e) - F(x.thr) for e in k.erev]
        self.tse = [F(x.thr) - F(e) for e in k.erev]
        self.act_dt = F(1) / F(k.act_tau)

    def xx1_gain_cor(self, v):
        x = self.k.xx1
        fact = (F(x.gain_cor_range) - (v / F(x.nvar))) / F(x.gain_cor_range)
        if fact < 0:
            g = F(x.gain) * v
            return g / (g + F(1))
        g = (F(x.gain) * (F(1) - F(x.gain_cor) * fact)) * v
        return g / (g + F(1))

    def noisy_xx1(self, v):
        x = self.k.xx1
        if v < 0:
            ex = -(v * self.sig_gain_nvar)
            if ex > F(50):
                return F(0)
            return self.sig_mult_eff / (F(1) + fast_exp32(ex))
        if v < F(x.interp_range):
            interp = F(1) - ((F(x.interp_range) - v) / F(x.interp_range))
            return self.sig_val_at0 + interp * self.interp_val
        return self.xx1_gain_cor(v)

    @staticmethod
    def fffb(p, ge_avg, ge_max, act_avg, fbi):
        if not p.on:
            return F(0), F(0)
        net = ge_avg + F(p.max_vs_avg) * (ge_max - ge_avg)
        ffi = F(p.ff) * (net - F(p.ff0)) if net > F(p.ff0) else F(0)
        fbi = fbi + (F(1) / F(p.fb_tau)) * (F(p.fb) * act_avg - fbi)
        return fbi, F(p.gi) * (ffi + fbi)

    def ge_thr(self, gi):
        g = self.k.gbar
        return (F(g[2]) * gi * self.est[2] + F(g[1]) * self.est[1]) / self.tse[0]

    def pool(self, raw):
        """KWTAPool on [d0, d1, d2, d3] with a fresh Inhibs slice; returns (act, cycles)"""
        k = self.k
        d0, d1, d2, d3 = raw.shape
        lay_n, pl_n = d0 * d1, d2 * d3
        ge = raw.reshape(lay_n, pl_n).astype(np.float32)
        act = ge.copy()
        s = F(0)
        for v in ge.ravel():
            s = s + v
        lay_ge_avg, lay_ge_max = s / F(ge.size), ge.max()
        p_avg, p_max = [], []
        for pi in range(lay_n):
            s = F(0)
            for v in ge[pi]:
                s = s + v
            p_avg.append(s / F(pl_n))
            p_max.append(ge[pi].max())
        lay_fbi, lay_act_avg = F(0), F(0)
        p_fbi, p_act = [F(0)] * lay_n, [F(0)] * lay_n
        cy = 0
        while cy < k.iters:
            lay_fbi, lay_gi = self.fffb(k.lay, lay_ge_avg, lay_ge_max, lay_act_avg, lay_fbi)
            lay_sum, max_del = F(0), F(0)
            for pi in range(lay_n):
                p_fbi[pi], gi = self.fffb(k.pool, p_avg[pi], p_max[pi], p_act[pi], p_fbi[pi])
                thr = self.ge_thr(max(lay_gi, gi))
                ps = F(0)
                for ui in range(pl_n):
                    nw = self.noisy_xx1(ge[pi, ui] * F(k.gbar[0]) - thr)
                    d = self.act_dt * (nw - act[pi, ui])
                    nw = act[pi, ui] + d
                    max_del = max(max_del, abs(d))
                    lay_sum = lay_sum + nw
                    ps = ps + nw
                    act[pi, ui] = nw
                p_act[pi] = ps / F(pl_n)
            lay_act_avg = lay_sum / F(ge.size)
            cy += 1
            if cy - 1 > 2 and max_del < F(k.del_act_thr):
                break
        return act.reshape(raw.shape), cy
