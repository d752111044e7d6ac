"""Device context and plan wrappers over the C ABI (one context per GPU, one process per GPU)."""
import ctypes as C

import numpy as np

from . import capi

_CTX = {}


def _check(lib, ctx, rc):
    if rc != capi.AUD_OK:
        msg = lib.aud_last_error(ctx).decode() if ctx else lib.aud_status_string(rc).decode()
        raise capi.AuditoryError(rc, msg or lib.aud_status_string(rc).decode())


class Context:
    """aud_ctx: owns the device, a stream for host-buffer calls and the RCCL communicator."""

    def __init__(self, device=0):
        self.lib = capi.load()
        h = C.c_void_p()
        rc = self.lib.aud_init(int(device), C.byref(h))
        if rc != capi.AUD_OK:
            raise capi.AuditoryError(rc, "aud_init(%d): no usable HIP device -- this library has no "
                                     "CPU fallback" % device)
        self.handle = h
        self.device = int(device)

    def check(self, rc):
        _check(self.lib, self.handle, rc)

    def pinned_empty(self, shape, dtype=np.float64):
        """a numpy array in pinned, device-visible host memory (aud_host_alloc): result tensors given to the _host / _sig
        calls through `out=` are written by the device itself -- no staging copy, no CPU widening pass.  The memory lives
        until pinned_free(array) or the context's close()."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self.check(self.lib.aud_host_alloc(self.handle, max(n, 16), C.byref(p)))
        buf = (C.c_char * max(n, 16)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.__array_interface__["data"][0]] = p.value
        return arr

    def pinned_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.__array_interface__["data"][0], None)
        if p is not None:
            self.check(self.lib.aud_host_free(self.handle, C.c_void_p(p)))

    def register_host(self, arr):
        """pin a numpy array the CALLER owns -- typically a mapping several processes share (np.memmap over /dev/shm) -- and
        make it device-visible (aud_host_register): result tensors that are views of it, given through `out=`, are written by
        the device like pinned_empty() ones.  One process per GPU, each passing its shard's slice of ONE [B, ...] float64
        tensor: the batch's features end in one host tensor without a collective.  unregister_host(arr) before unmapping."""
        self.check(self.lib.aud_host_register(self.handle, C.c_void_p(arr.__array_interface__["data"][0]), int(arr.nbytes)))

    def unregister_host(self, arr):
        self.check(self.lib.aud_host_unregister(self.handle, C.c_void_p(arr.__array_interface__["data"][0])))

    def close(self):
        if self.handle:
            self.lib.aud_shutdown(self.handle)
            self.handle = None


class Signal:
    """aud_signal: a signal kept resident on the device between calls (SndEnv.Signal: ProcessSegment runs once per segment
    on the same tensor).  samples: float64 (the Signal tensor), float32, or int16 PCM (normalised /0x7FFF on the device)."""

    _DT = {np.dtype(np.float64): capi.AUD_F64, np.dtype(np.float32): capi.AUD_F32, np.dtype(np.int16): capi.AUD_I16}

    def __init__(self, ctx, samples=None):
        """samples given: a SNAPSHOT of them (aud_signal_upload).  None: an empty handle for sync()."""
        self.ctx, self.n, self.handle = ctx, 0, C.c_void_p()
        self.uploaded_bytes = 0
        if samples is None:
            return
        samples = np.ascontiguousarray(samples)
        if samples.dtype not in self._DT:
            raise TypeError("samples must be float64, float32 or int16")
        self.n = int(samples.size)
        ctx.check(ctx.lib.aud_signal_upload(ctx.handle, samples.ctypes.data_as(C.c_void_p), self._DT[samples.dtype],
                                            samples.size, C.byref(self.handle)))
        self.uploaded_bytes = samples.nbytes

    def sync(self, samples):
        """aud_signal_sync: make the device copy EQUAL to `samples` -- compared byte for byte with the handle's host shadow of
        what the device holds; only a differing span (everything the first time) crosses the link.  Returns the bytes uploaded."""
        samples = np.ascontiguousarray(samples)
        if samples.dtype not in self._DT:
            raise TypeError("samples must be float64, float32 or int16")
        up = C.c_int64(0)
        self.ctx.check(self.ctx.lib.aud_signal_sync(self.ctx.handle, C.byref(self.handle), samples.ctypes.data_as(C.c_void_p),
                                                    self._DT[samples.dtype], samples.size, C.byref(up)))
        self.n, self.uploaded_bytes = int(samples.size), int(up.value)
        return self.uploaded_bytes

    def close(self):
        if self.handle:
            self.ctx.lib.aud_signal_destroy(self.handle)
            self.handle = None

    def __del__(self):   # (safe in any order with Context.close(): aud_shutdown detaches the handle, the destroy frees it)
        try:
            self.close()
        except Exception:
            pass


def get_ctx(device=0):
    if device not in _CTX:
        _CTX[device] = Context(device)
    return _CTX[device]


def _dptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Plan:
    """aud_plan: immutable tables (twiddles, mel triangles, gabor taps) resident on the device."""

    def __init__(self, ctx, win_samples, step_samples, segment_steps, border_steps, dft, fbank,
                 bin_pts, mel_filters, gabor_set=None, gabor_filters=None, compute_dtype=capi.AUD_F64,
                 mfcc_coefs=0):
        self.ctx = ctx
        self.lib = ctx.lib
        self.N, self.S, self.T, self.border = win_samples, step_samples, segment_steps, border_steps
        self.H = win_samples // 2 + 1
        self.nf = fbank.n_filters
        self.compute_dtype = compute_dtype
        bp = np.ascontiguousarray(bin_pts, np.int32)
        mf = np.ascontiguousarray(mel_filters, np.float64)
        d = capi.PlanDesc()
        d.win_samples, d.step_samples = win_samples, step_samples
        d.segment_steps, d.border_steps = segment_steps, border_steps
        d.dft, d.mel = dft, fbank
        self.n_gabor = 0
        gk_ptr = None
        if gabor_filters is not None and len(gabor_filters):
            gk = np.ascontiguousarray(gabor_filters, np.float64)
            d.n_gabor = gk.shape[0]
            d.gabor = gabor_set
            gk_ptr = _dptr(gk, C.c_double)
            self.n_gabor = gk.shape[0]
            self.gabor_set = gabor_set
        d.compute_dtype = compute_dtype
        d.mfcc_coefs = int(mfcc_coefs)
        self.mfcc_coefs = int(mfcc_coefs)
        h = C.c_void_p()
        ctx.check(self.lib.aud_plan_create(ctx.handle, C.byref(d), _dptr(bp, C.c_int32), _dptr(mf, C.c_double), gk_ptr,
                                           C.byref(h)))
        self.handle = h

    @property
    def kernel_name(self):
        return self.lib.aud_plan_kernel_name(self.handle).decode()

    def info(self, name):
        """aud_plan_get_info: "lds_bytes", "waves_per_wg", "wgs_per_cu", "frames_per_wave", "bluestein_L" of the mel kernel"""
        v = C.c_int64(0)
        self.ctx.check(self.lib.aud_plan_get_info(self.handle, name.encode(), C.byref(v)))
        return int(v.value)

    def set_option(self, name, value):
        """aud_plan_set_option: "kernel" (0 auto / 1 generic), "xcd_remap" (1 / 0)"""
        self.ctx.check(self.lib.aud_plan_set_option(self.handle, name.encode(), int(value)))

    # ---- device-pointer calls (ints are raw device addresses; stream is a hipStream_t) ----
    def melspec_dev(self, sig_ptr, sig_dtype, items_ptr, n_items, mel_ptr, power_ptr=0,
                    log_power_ptr=0, stream=0):
        self.ctx.check(self.lib.aud_melspec_batch_dev(self.handle, sig_ptr, sig_dtype, items_ptr,
                                                      n_items, mel_ptr, power_ptr or None,
                                                      log_power_ptr or None, stream or None))

    def gabor_dev(self, mel_ptr, n_items, rows, cols, out_shape, out_ptr, by_time=False, stream=0):
        shp = (C.c_int32 * len(out_shape))(*out_shape)
        self.ctx.check(self.lib.aud_gabor_batch_dev(self.handle, mel_ptr, n_items, rows, cols,
                                                    len(out_shape), shp, int(by_time), out_ptr,
                                                    stream or None))

    def process_dev(self, sig_ptr, sig_dtype, items_ptr, n_items, mel_ptr, pools_y, pools_x,
                    gabor_ptr, stream=0):
        self.ctx.check(self.lib.aud_process_batch_dev(self.handle, sig_ptr, sig_dtype, items_ptr,
                                                      n_items, mel_ptr, pools_y, pools_x, gabor_ptr,
                                                      stream or None))

    # ---- host-buffer calls (float64 in / float64+float32 out, like the Go tensors) ---------
    def segment_workspace_bytes(self, n_items):
        n = C.c_int64(0)
        self.ctx.check(self.lib.aud_segment_workspace_bytes(self.handle, n_items, C.byref(n)))
        return n.value

    def segment_dev(self, sig_ptr, sig_dtype, items_ptr, n_items, mel_ptr, power_ptr, log_power_ptr, mfcc_ptr,
                    deltas_ptr, delta_deltas_ptr, energy_ptr, workspace_ptr, workspace_bytes, stream=0):
        """SndEnv.ProcessSegment with Mel.MFCC on for n_items segments (aud_segment_batch_dev); 0 = NULL for the optional
        outputs"""
        self.ctx.check(self.lib.aud_segment_batch_dev(self.handle, sig_ptr, sig_dtype, items_ptr, n_items, mel_ptr,
                                                      power_ptr or None, log_power_ptr or None, mfcc_ptr,
                                                      deltas_ptr or None, delta_deltas_ptr or None, energy_ptr or None,
                                                      workspace_ptr, workspace_bytes, stream or None))

    def melspec_host(self, sig, items, want_power=False, want_log_power=False):
        sig = np.ascontiguousarray(sig, np.float64)
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        n = len(items)
        mel = np.zeros((n, self.nf, self.T), np.float64)
        power = np.zeros((n, self.H, self.T), np.float64) if want_power else None
        logp = np.zeros((n, self.H, self.T), np.float64) if want_log_power else None
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        self.ctx.check(self.lib.aud_melspec_batch_host(self.handle, vp(sig), sig.size, vp(items), n,
                                                       vp(mel), vp(power), vp(logp)))
        return mel, power, logp

    def melspec_sig(self, signal, items, want_power=False, want_log_power=False, out=None):
        """melspec_host on a resident Signal (aud_melspec_batch_sig): only the items go up, only the results come back.
        out = (mel, power or None, log_power or None): float64 result arrays to fill -- from Context.pinned_empty the device
        writes them directly"""
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        n = len(items)
        if out is not None:
            mel, power, logp = out
            assert mel.shape == (n, self.nf, self.T) and mel.dtype == np.float64 and mel.flags.c_contiguous
            for a in (power, logp):
                assert a is None or (a.shape == (n, self.H, self.T) and a.dtype == np.float64 and a.flags.c_contiguous)
        else:
            mel = np.zeros((n, self.nf, self.T), np.float64)
            power = np.zeros((n, self.H, self.T), np.float64) if want_power else None
            logp = np.zeros((n, self.H, self.T), np.float64) if want_log_power else None
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        self.ctx.check(self.lib.aud_melspec_batch_sig(self.handle, signal.handle, vp(items), n, vp(mel), vp(power), vp(logp)))
        return mel, power, logp

    def melspec_live(self, signal, sig, items, want_power=False, want_log_power=False, out=None):
        """aud_melspec_batch_live: melspec_host's result on the float64 array `sig` as it is NOW, from the resident copy
        `signal` (a runtime.Signal, created empty) -- the 4 KB blocks the items' frames read are compared with the copy's host
        shadow and uploaded where they differ, then the call runs on the device copy.  signal.uploaded_bytes says what moved."""
        sig = np.ascontiguousarray(sig, np.float64)
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        n = len(items)
        if out is not None:
            mel, power, logp = out
        else:
            mel = np.zeros((n, self.nf, self.T), np.float64)
            power = np.zeros((n, self.H, self.T), np.float64) if want_power else None
            logp = np.zeros((n, self.H, self.T), np.float64) if want_log_power else None
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        up = C.c_int64(0)
        self.ctx.check(self.lib.aud_melspec_batch_live(self.handle, C.byref(signal.handle), vp(sig), sig.size, vp(items), n, vp(mel),
                                                       vp(power), vp(logp), C.byref(up)))
        signal.n, signal.uploaded_bytes = int(sig.size), int(up.value)
        return mel, power, logp

    def melspec_mfcc_live(self, signal, sig, items, deltas=True):
        """aud_melspec_mfcc_batch_live: melspec_mfcc_host's result on `sig` as it is now, from the resident copy `signal`"""
        sig = np.ascontiguousarray(sig, np.float64)
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        n, nc = len(items), self.mfcc_coefs
        out = dict(mel=np.zeros((n, self.nf, self.T)), power=np.zeros((n, self.H, self.T)),
                   log_power=np.zeros((n, self.H, self.T)), mfcc=np.zeros((n, nc, self.T)),
                   deltas=np.zeros((n, nc, self.T)) if deltas else None,
                   delta_deltas=np.zeros((n, nc, self.T)) if deltas else None, energy=np.zeros((n, self.T)))
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        up = C.c_int64(0)
        self.ctx.check(self.lib.aud_melspec_mfcc_batch_live(
            self.handle, C.byref(signal.handle), vp(sig), sig.size, vp(items), n, vp(out["mel"]), vp(out["power"]),
            vp(out["log_power"]), vp(out["mfcc"]), vp(out["deltas"]), vp(out["delta_deltas"]), vp(out["energy"]), C.byref(up)))
        signal.n, signal.uploaded_bytes = int(sig.size), int(up.value)
        return out

    def melspec_mfcc_sig(self, signal, items, deltas=True):
        """melspec_mfcc_host on a resident Signal (aud_melspec_mfcc_batch_sig)"""
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        n, nc = len(items), self.mfcc_coefs
        out = dict(mel=np.zeros((n, self.nf, self.T)), power=np.zeros((n, self.H, self.T)),
                   log_power=np.zeros((n, self.H, self.T)), mfcc=np.zeros((n, nc, self.T)),
                   deltas=np.zeros((n, nc, self.T)) if deltas else None,
                   delta_deltas=np.zeros((n, nc, self.T)) if deltas else None, energy=np.zeros((n, self.T)))
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        self.ctx.check(self.lib.aud_melspec_mfcc_batch_sig(
            self.handle, signal.handle, vp(items), n, vp(out["mel"]), vp(out["power"]), vp(out["log_power"]),
            vp(out["mfcc"]), vp(out["deltas"]), vp(out["delta_deltas"]), vp(out["energy"])))
        return out

    def melspec_mfcc_host(self, sig, items, deltas=True):
        """ProcessSegment with Mel.MFCC on, for all items at once.  Returns a dict of float64 arrays:
        mel [n, nf, T], power / log_power [n, H, T], mfcc / deltas / delta_deltas [n, NCoefs, T], energy [n, T]"""
        sig = np.ascontiguousarray(sig, np.float64)
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        n, nc = len(items), self.mfcc_coefs
        out = dict(mel=np.zeros((n, self.nf, self.T)), power=np.zeros((n, self.H, self.T)),
                   log_power=np.zeros((n, self.H, self.T)), mfcc=np.zeros((n, nc, self.T)),
                   deltas=np.zeros((n, nc, self.T)) if deltas else None,
                   delta_deltas=np.zeros((n, nc, self.T)) if deltas else None, energy=np.zeros((n, self.T)))
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        self.ctx.check(self.lib.aud_melspec_mfcc_batch_host(
            self.handle, vp(sig), sig.size, vp(items), n, vp(out["mel"]), vp(out["power"]), vp(out["log_power"]),
            vp(out["mfcc"]), vp(out["deltas"]), vp(out["delta_deltas"]), vp(out["energy"])))
        return out

    def gabor_host(self, mel, out, by_time=False):
        """mel: f64 [n, rows, cols]; out: f32 [n, ...] modified in place"""
        mel = np.ascontiguousarray(mel, np.float64)
        assert out.dtype == np.float32 and out.flags.c_contiguous
        n, rows, cols = mel.shape
        shp = (C.c_int32 * (out.ndim - 1))(*out.shape[1:])
        rc = self.lib.aud_gabor_batch_host(self.handle, mel.ctypes.data_as(C.c_void_p), n, rows,
                                           cols, out.ndim - 1, shp, int(by_time),
                                           out.ctypes.data_as(C.c_void_p))
        self.ctx.check(rc)
        return out

    def close(self):
        if self.handle:
            self.lib.aud_plan_destroy(self.handle)
            self.handle = None


ITEM_DTYPE = np.dtype([("sig_off", np.int64), ("sig_len", np.int32), ("start0", np.int32),
                       ("sig_stride", np.int32), ("reserved", np.int32)])
assert ITEM_DTYPE.itemsize == C.sizeof(capi.Item)


def make_items(sig_off, sig_len, start0, sig_stride=1):
    """aud_item array; sig_stride = 2 with sig_off = 0 / 1 addresses the channels of interleaved stereo PCM"""
    n = len(sig_off)
    it = np.zeros(n, ITEM_DTYPE)
    it["sig_off"], it["sig_len"], it["start0"], it["sig_stride"] = sig_off, sig_len, start0, sig_stride
    return it
