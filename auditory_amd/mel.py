"""Host-side mirror of the reference package `mel` (mel/mel.go): parameters and table setup.
The per-frame arithmetic (FilterDft) runs on the GPU inside the fused batch kernel."""
import ctypes as C

import numpy as np

from . import capi


def FreqToMel(freq):
    """mel/mel.go:156-158"""
    return capi.load().aud_freq_to_mel(float(freq))


def MelToFreq(mel):
    """mel/mel.go:161-163"""
    return capi.load().aud_mel_to_freq(float(mel))


def FreqToBin(freq, nFft, sampleRate):
    """mel/mel.go:166-168"""
    return capi.load().aud_freq_to_bin(float(freq), float(nFft), float(sampleRate))


class FilterBank:
    """mel.FilterBank, mel/mel.go:16-44"""

    def __init__(self):
        self.NFilters = 0
        self.LoHz = 0.0
        self.HiHz = 0.0
        self.LogOff = 0.0
        self.LogMin = 0.0
        self.Renorm = False
        self.RenormMin = 0.0
        self.RenormMax = 0.0
        self.RenormScale = 0.0

    def Defaults(self):
        """mel/mel.go:171-180"""
        c = capi.MelFBank()
        capi.load().aud_mel_defaults(c)
        self.from_c(c)

    def to_c(self):
        return capi.MelFBank(self.NFilters, self.LoHz, self.HiHz, self.LogOff, self.LogMin,
                             int(self.Renorm), self.RenormMin, self.RenormMax, self.RenormScale)

    def from_c(self, c):
        self.NFilters, self.LoHz, self.HiHz = c.n_filters, c.lo_hz, c.hi_hz
        self.LogOff, self.LogMin, self.Renorm = c.log_off, c.log_min, bool(c.renorm)
        self.RenormMin, self.RenormMax, self.RenormScale = c.renorm_min, c.renorm_max, c.renorm_scale


class Params:
    """mel.Params, mel/mel.go:47-66"""

    def __init__(self):
        self.FBank = FilterBank()
        self.BinPts = None
        self.HzPts = None
        self.MFCC = False
        self.Deltas = False
        self.NCoefs = 0

    def Defaults(self):
        """mel/mel.go:69-74"""
        self.FBank.Defaults()
        self.MFCC = True
        self.NCoefs = 13
        self.Deltas = True

    def FilterDft(self, step, dftPowerOut, segmentData, fBankData, filters, plan):
        """mel/mel.go:120-153 for ONE step on the GPU (see dft.Params.Filter about per-frame launches).
        `filters` is accepted for signature parity; the plan already holds the table on the device."""
        import ctypes as C
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        pw = np.ascontiguousarray(dftPowerOut, np.float64)
        plan.ctx.check(plan.lib.aud_mel_filter_dft_host(plan.handle, int(step), vp(pw), vp(segmentData), vp(fBankData)))

    def FftReal(self, out, inp):
        """mel/mel.go:183-189: out[i] = complex(inp[i], 0) -- a host copy, no arithmetic"""
        out[:] = 0
        out.real[:] = inp[:len(out)]

    def CepstrumDct(self, step, fBankData, mfccSegment, mfccDct, plan):
        """mel/mel.go:192-212 for ONE step on the GPU: DCT-I of the nf log-mel values, c0 <- ln(1 + c0^2), the
        first NCoefs into column `step` of mfccSegment [NCoefs, T]; mfccDct [nf] ends as a copy of fBankData.
        `plan` must have been created with mfcc_coefs = NCoefs."""
        import ctypes as C
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        fb = np.ascontiguousarray(fBankData, np.float64)
        plan.ctx.check(plan.lib.aud_cepstrum_dct_host(plan.handle, int(step), vp(fb), vp(mfccSegment), vp(mfccDct)))

    def InitFilters(self, dftSize, sampleRate):
        """mel/mel.go:77-117.  Returns the [NFilters, NFilters+2] float64 filter tensor (the Go
        code fills the tensor passed by the caller); sets BinPts / HzPts and clears FBank.Renorm."""
        nf = self.FBank.NFilters
        c = self.FBank.to_c()
        bins = np.zeros(nf + 2, np.int32)
        hz = np.zeros(nf + 2, np.float64)
        filt = np.zeros((nf, nf + 2), np.float64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = capi.load().aud_mel_init_filters(c, int(dftSize), int(sampleRate), vp(bins), vp(hz),
                                              vp(filt))
        if rc != capi.AUD_OK:
            raise capi.AuditoryError(rc, "InitFilters: a triangle runs past the end of the filter "
                                     "tensor (the Go code panics here)")
        self.FBank.from_c(c)
        self.BinPts, self.HzPts = bins, hz
        return filt
