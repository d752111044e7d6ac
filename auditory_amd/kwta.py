"""Host-side mirror of the k-WTA stage SndEnv.ApplyKwta runs on the gabor output
(sound/sndenv.go:313-323): kwta.KWTA of github.com/emer/vision v1.1.15 (kwta/kwta.go) with the
fffb.Params / nxx1.Params / chans.Chans of github.com/emer/leabra v1.1.48 it is built from.

Field names follow those Go types.  The settling itself runs on the GPU (aud_kwta_batch_*, kwta.hip);
there is no CPU path here.  NeighInhib is not built (the external-inhibition tensor is zeros, as in the
reference whenever NeighInhib.On is false)."""
import ctypes as C

import numpy as np

from . import capi, runtime


class FFFBParams:
    """leabra fffb.Params"""

    def __init__(self, c):
        self._c = c

    On = property(lambda s: bool(s._c.on), lambda s, v: setattr(s._c, "on", int(bool(v))))
    Gi = property(lambda s: s._c.gi, lambda s, v: setattr(s._c, "gi", v))
    FF = property(lambda s: s._c.ff, lambda s, v: setattr(s._c, "ff", v))
    FB = property(lambda s: s._c.fb, lambda s, v: setattr(s._c, "fb", v))
    FBTau = property(lambda s: s._c.fb_tau, lambda s, v: setattr(s._c, "fb_tau", v))
    MaxVsAvg = property(lambda s: s._c.max_vs_avg, lambda s, v: setattr(s._c, "max_vs_avg", v))
    FF0 = property(lambda s: s._c.ff0, lambda s, v: setattr(s._c, "ff0", v))


class XX1Params:
    """leabra nxx1.Params (settable fields)"""

    def __init__(self, c):
        self._c = c

    Thr = property(lambda s: s._c.thr, lambda s, v: setattr(s._c, "thr", v))
    Gain = property(lambda s: s._c.gain, lambda s, v: setattr(s._c, "gain", v))
    NVar = property(lambda s: s._c.nvar, lambda s, v: setattr(s._c, "nvar", v))
    VmActThr = property(lambda s: s._c.vm_act_thr, lambda s, v: setattr(s._c, "vm_act_thr", v))
    SigMult = property(lambda s: s._c.sig_mult, lambda s, v: setattr(s._c, "sig_mult", v))
    SigMultPow = property(lambda s: s._c.sig_mult_pow, lambda s, v: setattr(s._c, "sig_mult_pow", v))
    SigGain = property(lambda s: s._c.sig_gain, lambda s, v: setattr(s._c, "sig_gain", v))
    InterpRange = property(lambda s: s._c.interp_range, lambda s, v: setattr(s._c, "interp_range", v))
    GainCorRange = property(lambda s: s._c.gain_cor_range, lambda s, v: setattr(s._c, "gain_cor_range", v))
    GainCor = property(lambda s: s._c.gain_cor, lambda s, v: setattr(s._c, "gain_cor", v))


class Chans:
    """leabra chans.Chans: E, L, I, K"""

    def __init__(self, arr):
        self._a = arr

    E = property(lambda s: s._a[0], lambda s, v: s._a.__setitem__(0, v))
    L = property(lambda s: s._a[1], lambda s, v: s._a.__setitem__(1, v))
    I = property(lambda s: s._a[2], lambda s, v: s._a.__setitem__(2, v))  # noqa: E741
    K = property(lambda s: s._a[3], lambda s, v: s._a.__setitem__(3, v))

    def SetAll(self, e, l, i, k):  # noqa: E741
        self._a[0], self._a[1], self._a[2], self._a[3] = e, l, i, k


class KWTA:
    """kwta.KWTA: On, Iters, DelActThr, LayFFFB, PoolFFFB, XX1, ActTau, Gbar, Erev"""

    def __init__(self):
        self.c = capi.KwtaParams()
        self.LayFFFB = FFFBParams(self.c.lay_fffb)
        self.PoolFFFB = FFFBParams(self.c.pool_fffb)
        self.XX1 = XX1Params(self.c.xx1)
        self.Gbar = Chans(self.c.gbar)
        self.Erev = Chans(self.c.erev)  # all zeros, like the Go zero value, until Defaults()

    On = property(lambda s: bool(s.c.on), lambda s, v: setattr(s.c, "on", int(bool(v))))
    Iters = property(lambda s: s.c.iters, lambda s, v: setattr(s.c, "iters", int(v)))
    DelActThr = property(lambda s: s.c.del_act_thr, lambda s, v: setattr(s.c, "del_act_thr", v))
    ActTau = property(lambda s: s.c.act_tau, lambda s, v: setattr(s.c, "act_tau", v))

    def Defaults(self):
        capi.load().aud_kwta_defaults(C.byref(self.c))

    def Update(self):
        """derived values are recomputed inside every call; kept for source compatibility"""

    # ---- the two entry points, on host tensors (one tensor = one item) ---------------------------
    def KWTAPool(self, raw, act, inhib=None, extGi=None, device=0, sum_order=0):
        """raw, act: float32 [PY, PX, d2, d3]; act holds the starting activations and receives the result
        (ApplyKwta copies raw into it first).  inhib: float32 [PY*PX, 2] carried pool state or None."""
        return _run_host(self, raw, act, True, inhib, extGi, device, sum_order)

    def KWTALayer(self, raw, act, extGi=None, device=0, sum_order=0):
        return _run_host(self, raw, act, False, None, extGi, device, sum_order)


def _run_host(k, raw, act, pool, inhib, extGi, device, sum_order):
    if extGi is not None and np.any(np.asarray(extGi) != 0):
        raise capi.AuditoryError(capi.AUD_EINVAL, "non-zero external inhibition (NeighInhib) is not built")
    raw = np.ascontiguousarray(raw, np.float32)
    if act.dtype != np.float32 or not act.flags.c_contiguous or act.shape != raw.shape:
        raise capi.AuditoryError(capi.AUD_EINVAL, "act must be a C-contiguous float32 array shaped like raw")
    shape = list(raw.shape) if pool else [raw.size, 1, 1, 1]
    if pool and raw.ndim != 4:
        raise capi.AuditoryError(capi.AUD_EINVAL, "KWTAPool needs a 4-D tensor")
    if inhib is not None and (inhib.dtype != np.float32 or inhib.shape != (shape[0] * shape[1], 2)
                              or not inhib.flags.c_contiguous):
        raise capi.AuditoryError(capi.AUD_EINVAL, "inhib must be float32 [PY*PX, 2]")
    ctx = runtime.get_ctx(device)
    cyc = np.zeros(1, np.int32)
    ctx.check(ctx.lib.aud_kwta_batch_host(
        ctx.handle, C.byref(k.c), raw.ctypes.data, act.ctypes.data, 1, *shape, int(pool), 0,
        inhib.ctypes.data if inhib is not None else None, int(sum_order), cyc.ctypes.data))
    return int(cyc[0])


def kwta_batch_host(k, raw, pool=True, state=None, device=0, sum_order=0):
    """Batch form on host memory: raw float32 [n_items, d0, d1, d2, d3] -> (act, cycles); settling
    starts from act = raw.  state float32 [n_items, d0*d1, 2] is updated in place when given."""
    raw = np.ascontiguousarray(raw, np.float32)
    if raw.ndim != 5:
        raise capi.AuditoryError(capi.AUD_EINVAL, "raw must be [n_items, d0, d1, d2, d3]")
    n = raw.shape[0]
    if state is not None and (state.dtype != np.float32 or not state.flags.c_contiguous
                              or state.shape != (n, raw.shape[1] * raw.shape[2], 2)):
        raise capi.AuditoryError(capi.AUD_EINVAL, "state must be float32 [n_items, d0*d1, 2]")
    act = np.empty_like(raw)
    cyc = np.zeros(n, np.int32)
    ctx = runtime.get_ctx(device)
    ctx.check(ctx.lib.aud_kwta_batch_host(
        ctx.handle, C.byref(k.c), raw.ctypes.data, act.ctypes.data, n, *raw.shape[1:], int(pool), 1,
        state.ctypes.data if state is not None else None, int(sum_order), cyc.ctypes.data))
    return act, cyc


def kwta_batch_dev(k, raw, act, pool=True, state=None, cycles=None, device=0, sum_order=0, stream=0,
                   start_from_raw=True):
    """Device-resident form: raw / act / state / cycles are objects with data_ptr() (torch tensors);
    raw float32 [n_items, d0, d1, d2, d3]."""
    ctx = runtime.get_ctx(device)
    n, d0, d1, d2, d3 = [int(v) for v in raw.shape]
    ctx.check(ctx.lib.aud_kwta_batch_dev(
        ctx.handle, C.byref(k.c), raw.data_ptr(), act.data_ptr(), n, d0, d1, d2, d3, int(pool),
        int(bool(start_from_raw)), state.data_ptr() if state is not None else None, int(sum_order),
        cycles.data_ptr() if cycles is not None else None, stream))
