"""Host-side mirror of the reference package `agabor` (agabor/gabor.go).
ToTensor runs on the host (it is per-plan setup); Convolve runs on the GPU."""
import ctypes as C

import numpy as np

from . import capi


class Filter:
    """agabor.Filter, agabor/gabor.go:17-42"""

    def __init__(self, WaveLen=0.0, Orientation=0.0, SigmaWidth=0.0, SigmaLength=0.0,
                 PhaseOffset=0.0, CircleEdge=False, Circular=False, Off=False):
        self.Off = Off
        self.WaveLen = WaveLen
        self.Orientation = Orientation
        self.SigmaWidth = SigmaWidth
        self.SigmaLength = SigmaLength
        self.PhaseOffset = PhaseOffset
        self.CircleEdge = CircleEdge
        self.Circular = Circular

    def to_c(self):
        return capi.GaborSpec(int(self.Off), self.WaveLen, self.Orientation, self.SigmaWidth,
                              self.SigmaLength, self.PhaseOffset, int(self.CircleEdge),
                              int(self.Circular))


class FilterSet:
    """agabor.FilterSet, agabor/gabor.go:45-70"""

    def __init__(self):
        self.SizeX = 0
        self.SizeY = 0
        self.StrideX = 0
        self.StrideY = 0
        self.Gain = 0.0
        self.Distribute = False
        self.Filters = np.zeros((0, 0, 0), np.float64)

    def to_c(self):
        return capi.GaborSet(self.SizeX, self.SizeY, self.StrideX, self.StrideY, self.Gain,
                             int(self.Distribute))


def Active(specs):
    """agabor/gabor.go:329-336"""
    return [s for s in specs if not s.Off]


def ToTensor(specs, fset):
    """agabor/gabor.go:89-222: renders the active specs into fset.Filters [n, SizeY, SizeX]"""
    arr = (capi.GaborSpec * max(len(specs), 1))()
    for i, s in enumerate(specs):
        arr[i] = s.to_c()
    n_act = len(Active(specs))
    out = np.zeros((n_act, fset.SizeY, fset.SizeX), np.float64)
    n_out = C.c_int(0)
    cset = fset.to_c()
    rc = capi.load().aud_gabor_to_tensor(arr, len(specs), cset, out.ctypes.data_as(C.c_void_p),
                                         C.byref(n_out))
    if rc != capi.AUD_OK:
        raise capi.AuditoryError(rc, "ToTensor")
    assert n_out.value == n_act
    fset.Filters = out


def Convolve(melData, filters, rawOut, byTime, plan=None):
    """agabor/gabor.go:225-315 on the GPU.  melData: float64 [rows, cols]; rawOut: float32 rank 2
    or 4, written in place.  Like the reference, a rejected shape logs and returns with rawOut
    untouched.  `plan` may carry a reusable device plan (runtime.Plan with these filters)."""
    from . import runtime
    melData = np.ascontiguousarray(melData, np.float64)
    own = plan is None
    if own:
        plan = _gabor_only_plan(filters)
    try:
        try:
            plan.gabor_host(melData[None], rawOut.reshape((1,) + rawOut.shape), byTime)
        except capi.AuditoryError as e:
            if e.status != capi.AUD_EINVAL:
                raise
            print("agabor.Convolve:", e)     # the reference logs and returns
    finally:
        if own:
            plan.close()


def _gabor_only_plan(filters, device=0):
    """a plan whose mel stage is a 1-filter dummy, for callers that only use Convolve"""
    from . import runtime
    fb = capi.MelFBank()
    fb.n_filters = 1
    dftp = capi.DftParams(1, -100.0, 1.0, 0.0, 1.0)
    return runtime.Plan(runtime.get_ctx(device), 4, 1, 1, 0, dftp, fb, np.zeros(3, np.int32),
                        np.zeros((1, 3)), filters.to_c(), filters.Filters)
