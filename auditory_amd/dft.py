"""Host-side mirror of the reference package `dft` (dft/dft.go)."""
from . import capi


class Params:
    """dft.Params, dft/dft.go:15-31"""

    def __init__(self):
        self.CompLogPow = False
        self.LogMin = 0.0
        self.LogOffSet = 0.0
        self.PrevSmooth = 0.0
        self.CurSmooth = 0.0

    def Defaults(self):
        """dft/dft.go:33-39"""
        c = capi.DftParams()
        c.prev_smooth = self.PrevSmooth
        capi.load().aud_dft_defaults(c)
        self.PrevSmooth, self.CurSmooth = c.prev_smooth, c.cur_smooth
        self.CompLogPow, self.LogOffSet, self.LogMin = bool(c.comp_log_pow), c.log_offset, c.log_min

    def Filter(self, step, windowIn, winSamples, power, logPower, powerForSegment, logPowerForSegment, plan):
        """dft/dft.go:42-85 for ONE step on the GPU (one launch per frame: correct, never fast -- batch at
        ProcessSegment level).  Arrays are float64 and written in place like the Go tensors; `power` is the
        previous step's carry on entry.  `plan` is the runtime.Plan of these parameters (SndEnv keeps one)."""
        import ctypes as C
        import numpy as np
        w = np.ascontiguousarray(windowIn, np.float64)
        assert len(w) == winSamples == plan.N
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        plan.ctx.check(plan.lib.aud_dft_filter_host(plan.handle, int(step), vp(w), vp(power), vp(logPower),
                                                    vp(powerForSegment), vp(logPowerForSegment)))

    def FftReal(self, fftCoefs, windowIn):
        """dft/dft.go:53-59: fftCoefs[i] = complex(windowIn[i], 0) -- a host copy, no arithmetic"""
        n = len(fftCoefs)
        fftCoefs[:] = 0
        fftCoefs.real[:] = windowIn[:n]

    def Power(self, step, winSamples, fftCoefs, power, logPower, powerForSegment, logPowerForSegment, plan):
        """dft/dft.go:62-85 for ONE step on coefficients the caller computed (complex128 [>= winSamples/2 + 1]);
        the squares, the PrevSmooth blend with the `power` carry and the log run on the GPU like Filter's."""
        import ctypes as C
        import numpy as np
        H = winSamples // 2 + 1
        assert H == plan.H
        co = np.ascontiguousarray(np.asarray(fftCoefs, np.complex128)[:H])
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        plan.ctx.check(plan.lib.aud_dft_power_host(plan.handle, int(step), vp(co), vp(power), vp(logPower),
                                                   vp(powerForSegment), vp(logPowerForSegment)))

    def to_c(self):
        return capi.DftParams(int(self.CompLogPow), self.LogMin, self.LogOffSet, self.PrevSmooth,
                              self.CurSmooth)
