"""ctypes binding of libauditory_hip.so (include/auditory_hip.h).

The library is the product; this file only declares its C ABI to Python.  There is no
fallback: if the shared library is missing, `load()` raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libauditory_hip.so")

AUD_OK, AUD_EINVAL, AUD_EHIP, AUD_ERCCL, AUD_ENOMEM, AUD_ESHORT, AUD_EBROKEN = 0, 1, 2, 3, 4, 5, 6
AUD_RESIDENT_AUTO_BYTES = 8 << 20   # include/auditory_hip.h
AUD_F64, AUD_F32, AUD_I16 = 0, 1, 2   # AUD_F64 == 0: a zeroed PlanDesc is the float64 (conforming) plan
AUD_FAST_F32 = AUD_F32                # compute_dtype opt-in: the float32 kernels (never a default)


class SoundParams(C.Structure):
    _fields_ = [("win_ms", C.c_double), ("step_ms", C.c_double), ("segment_ms", C.c_double),
                ("stride_ms", C.c_double), ("border_steps", C.c_int32), ("channel", C.c_int32),
                ("win_samples", C.c_int32), ("step_samples", C.c_int32),
                ("segment_samples", C.c_int32), ("stride_samples", C.c_int32),
                ("segment_steps", C.c_int32)]


class DftParams(C.Structure):
    _fields_ = [("comp_log_pow", C.c_int32), ("log_min", C.c_double), ("log_offset", C.c_double),
                ("prev_smooth", C.c_double), ("cur_smooth", C.c_double)]


class MelFBank(C.Structure):
    _fields_ = [("n_filters", C.c_int32), ("lo_hz", C.c_double), ("hi_hz", C.c_double),
                ("log_off", C.c_double), ("log_min", C.c_double), ("renorm", C.c_int32),
                ("renorm_min", C.c_double), ("renorm_max", C.c_double),
                ("renorm_scale", C.c_double)]


class GaborSpec(C.Structure):
    _fields_ = [("off", C.c_int32), ("wave_len", C.c_double), ("orientation", C.c_double),
                ("sigma_width", C.c_double), ("sigma_length", C.c_double),
                ("phase_offset", C.c_double), ("circle_edge", C.c_int32), ("circular", C.c_int32)]


class GaborSet(C.Structure):
    _fields_ = [("size_x", C.c_int32), ("size_y", C.c_int32), ("stride_x", C.c_int32),
                ("stride_y", C.c_int32), ("gain", C.c_double), ("distribute", C.c_int32)]


class Item(C.Structure):
    _fields_ = [("sig_off", C.c_int64), ("sig_len", C.c_int32), ("start0", C.c_int32),
                ("sig_stride", C.c_int32), ("reserved", C.c_int32)]


class PlanDesc(C.Structure):
    _fields_ = [("win_samples", C.c_int32), ("step_samples", C.c_int32),
                ("segment_steps", C.c_int32), ("border_steps", C.c_int32),
                ("dft", DftParams), ("mel", MelFBank),
                ("n_gabor", C.c_int32), ("gabor", GaborSet), ("compute_dtype", C.c_int32),
                ("mfcc_coefs", C.c_int32)]


class FffbParams(C.Structure):
    _fields_ = [("on", C.c_int32)] + [(n, C.c_float) for n in ("gi", "ff", "fb", "fb_tau", "max_vs_avg", "ff0")]


class Nxx1Params(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("thr", "gain", "nvar", "vm_act_thr", "sig_mult", "sig_mult_pow",
                                         "sig_gain", "interp_range", "gain_cor_range", "gain_cor")]


class KwtaParams(C.Structure):
    _fields_ = [("on", C.c_int32), ("iters", C.c_int32), ("del_act_thr", C.c_float),
                ("lay_fffb", FffbParams), ("pool_fffb", FffbParams), ("xx1", Nxx1Params),
                ("act_tau", C.c_float), ("gbar", C.c_float * 4), ("erev", C.c_float * 4)]


# every symbol include/auditory_hip.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "aud_version": (C.c_int, []),
    "aud_status_string": (C.c_char_p, [C.c_int]),
    "aud_msec_to_samples": (C.c_int, [C.c_double, C.c_int]),
    "aud_samples_to_msec": (C.c_double, [C.c_int, C.c_int]),
    "aud_sound_params_defaults": (None, [C.POINTER(SoundParams)]),
    "aud_sound_params_derive": (C.c_int, [C.POINTER(SoundParams), C.c_int]),
    "aud_seg_cnt": (C.c_int, [C.c_int] * 4),
    "aud_tail": (C.c_int, [C.c_int] * 3),
    "aud_pad_len": (C.c_int, [C.c_int] * 4),
    "aud_pcm_to_float": (C.c_double, [C.c_int, C.c_int]),
    "aud_adjust_for_silence": (C.c_int, [C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int)]),
    "aud_dft_defaults": (None, [C.POINTER(DftParams)]),
    "aud_mel_defaults": (None, [C.POINTER(MelFBank)]),
    "aud_freq_to_mel": (C.c_double, [C.c_double]),
    "aud_mel_to_freq": (C.c_double, [C.c_double]),
    "aud_freq_to_bin": (C.c_int, [C.c_double] * 3),
    "aud_mel_init_filters": (C.c_int, [C.POINTER(MelFBank), C.c_int, C.c_int, _VP, _VP, _VP]),
    "aud_gabor_active": (C.c_int, [C.POINTER(GaborSpec), C.c_int, C.POINTER(GaborSpec)]),
    "aud_gabor_to_tensor": (C.c_int, [C.POINTER(GaborSpec), C.c_int, C.POINTER(GaborSet), _VP,
                                      C.POINTER(C.c_int)]),
    "aud_gabor_iter_space": (C.c_int, [C.POINTER(GaborSet), C.c_int, C.c_int, C.c_int, _VP,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32)]),
    "aud_init": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "aud_shutdown": (C.c_int, [_VP]),
    "aud_last_error": (C.c_char_p, [_VP]),
    "aud_device_id": (C.c_int, [_VP]),
    "aud_plan_create": (C.c_int, [_VP, C.POINTER(PlanDesc), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(_VP)]),
    "aud_plan_destroy": (C.c_int, [_VP]),
    "aud_plan_kernel_name": (C.c_char_p, [_VP]),
    "aud_plan_set_option": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "aud_plan_get_info": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int64)]),
    "aud_melspec_batch_dev": (C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, _VP, _VP, _VP, _VP]),
    "aud_mfcc_batch_dev": (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "aud_segment_workspace_bytes": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int64)]),
    "aud_segment_batch_dev": (C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP]),
    "aud_melspec_mfcc_batch_host": (C.c_int, [_VP, _VP, C.c_int64, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "aud_gabor_batch_dev": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP, C.c_int,
                                      _VP, _VP]),
    "aud_process_batch_dev": (C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, _VP, C.c_int, C.c_int,
                                        _VP, _VP]),
    "aud_melspec_batch_host": (C.c_int, [_VP, _VP, C.c_int64, _VP, C.c_int, _VP, _VP, _VP]),
    "aud_host_alloc": (C.c_int, [_VP, C.c_int64, C.POINTER(C.c_void_p)]),
    "aud_host_free": (C.c_int, [_VP, _VP]),
    "aud_host_register": (C.c_int, [_VP, _VP, C.c_int64]),
    "aud_host_unregister": (C.c_int, [_VP, _VP]),
    "aud_signal_upload": (C.c_int, [_VP, _VP, C.c_int, C.c_int64, C.POINTER(C.c_void_p)]),
    "aud_signal_sync": (C.c_int, [_VP, C.POINTER(C.c_void_p), _VP, C.c_int, C.c_int64, C.POINTER(C.c_int64)]),
    "aud_signal_destroy": (C.c_int, [_VP]),
    "aud_signal_len": (C.c_int64, [_VP]),
    "aud_melspec_batch_sig": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, _VP, _VP]),
    "aud_melspec_mfcc_batch_sig": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "aud_melspec_batch_live": (C.c_int, [_VP, C.POINTER(C.c_void_p), _VP, C.c_int64, _VP, C.c_int, _VP, _VP, _VP, C.POINTER(C.c_int64)]),
    "aud_melspec_mfcc_batch_live": (C.c_int, [_VP, C.POINTER(C.c_void_p), _VP, C.c_int64, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                            C.POINTER(C.c_int64)]),
    "aud_snd_to_window": (C.c_int, [_VP, C.c_int64, C.c_int64, C.c_int, _VP]),
    "aud_dft_filter_host": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, _VP, _VP]),
    "aud_mel_filter_dft_host": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP]),
    "aud_dft_power_host": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, _VP, _VP]),
    "aud_cepstrum_dct_host": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP]),
    "aud_gabor_batch_host": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP, C.c_int,
                                       _VP]),
    "aud_kwta_defaults": (None, [C.POINTER(KwtaParams)]),
    "aud_kwta_batch_dev": (C.c_int, [_VP, C.POINTER(KwtaParams), _VP, _VP] + [C.c_int] * 7 + [_VP, C.c_int, _VP, _VP]),
    "aud_kwta_batch_host": (C.c_int, [_VP, C.POINTER(KwtaParams), _VP, _VP] + [C.c_int] * 7 + [_VP, C.c_int, _VP]),
    "aud_comm_unique_id": (C.c_int, [_VP]),
    "aud_comm_init": (C.c_int, [_VP, C.c_int, C.c_int, _VP]),
    "aud_comm_destroy": (C.c_int, [_VP]),
    "aud_allgather_dev": (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP]),
    "aud_gather_create": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int64, C.POINTER(_VP), _VP]),
    "aud_gather_open_peer": (C.c_int, [_VP, C.c_int, _VP]),
    "aud_allgather_direct_dev": (C.c_int, [_VP, _VP, C.c_int64, C.POINTER(C.c_int), _VP]),
    "aud_gather_wait_dev": (C.c_int, [_VP, _VP]),
    "aud_gather_timeouts": (C.c_int, [_VP, C.POINTER(C.c_int)]),
    "aud_gather_flags_fine": (C.c_int, [_VP]),
    "aud_gather_destroy": (C.c_int, [_VP]),
}

_LIB = None


class AuditoryError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__("auditory_hip status %d%s" % (status, (": " + msg) if msg else ""))


def load():
    """dlopen libauditory_hip.so and type every entry point.  Raises if it is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libauditory_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `python -m auditory_amd.build`; there is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB
