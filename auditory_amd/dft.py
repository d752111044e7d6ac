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

    def to_c(self):
        return capi.DftParams(int(self.CompLogPow), self.LogMin, self.LogOffSet, self.PrevSmooth,
                              self.CurSmooth)
