"""Error behaviour of the C ABI (emulator build, CPU): every misuse returns a status code and a message,
writes nothing, and never crashes -- the contract the cgo shim relies on."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import workloads as W
from auditory_amd import capi, runtime

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emul"))


@pytest.fixture(scope="module")
def emu():
    import backend
    with backend.emulated("plain") as lib:
        yield lib


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_melspec_argument_errors(orc, emu):
    oc = W.OracleCfg(orc, "sndenv_16k_n400_nf32")
    plan = W.product_plan(oc)
    lib, h = plan.lib, plan.handle
    sig = np.zeros(4000)
    items = runtime.make_items([0], [4000], [0])
    mel = np.full((1, 32, 14), 7.0)
    # null buffers
    assert lib.aud_melspec_batch_host(h, None, 4000, _vp(items), 1, _vp(mel), None, None) == capi.AUD_EINVAL
    assert lib.aud_melspec_batch_host(h, _vp(sig), 4000, None, 1, _vp(mel), None, None) == capi.AUD_EINVAL
    assert lib.aud_melspec_batch_host(h, _vp(sig), 4000, _vp(items), 1, None, None, None) == capi.AUD_EINVAL
    assert lib.aud_melspec_batch_host(h, _vp(sig), 4000, _vp(items), -1, _vp(mel), None, None) == capi.AUD_EINVAL
    # item outside the signal buffer
    bad = runtime.make_items([100], [4000], [0])
    assert lib.aud_melspec_batch_host(h, _vp(sig), 4000, _vp(bad), 1, _vp(mel), None, None) == capi.AUD_EINVAL
    assert b"outside" in lib.aud_last_error(plan.ctx.handle)
    assert (mel == 7.0).all()                                   # nothing was written
    # bad sample type on the device entry point
    assert lib.aud_melspec_batch_dev(h, _vp(sig), 9, _vp(items), 1, _vp(mel), None, None, None) == capi.AUD_EINVAL
    # null plan
    assert lib.aud_melspec_batch_host(None, _vp(sig), 4000, _vp(items), 1, _vp(mel), None, None) == capi.AUD_EINVAL
    # MFCC entry points on a plan created without mfcc_coefs
    assert lib.aud_mfcc_batch_dev(h, _vp(items), 1, _vp(mel), _vp(mel), _vp(mel), None, None, None, None) == capi.AUD_EINVAL
    # gabor on a plan without gabor filters
    out = np.ones((1, 8, 2, 2, 8), np.float32)
    shp = (C.c_int32 * 4)(8, 2, 2, 8)
    assert lib.aud_gabor_batch_host(h, _vp(mel), 1, 32, 14, 4, shp, 0, _vp(out)) == capi.AUD_EINVAL
    assert (out == 1).all()
    plan.close()


def test_plan_argument_errors(orc, emu):
    oc = W.OracleCfg(orc, "sndenv_16k_n400_nf32")
    ctx = runtime.get_ctx(0)
    lib = ctx.lib
    d = capi.PlanDesc()
    h = C.c_void_p()
    assert lib.aud_plan_create(ctx.handle, None, None, None, None, C.byref(h)) == capi.AUD_EINVAL
    assert lib.aud_plan_create(None, C.byref(d), None, None, None, C.byref(h)) == capi.AUD_EINVAL
    d.win_samples, d.step_samples, d.segment_steps = 400, 160, 14
    d.mel.n_filters = 32
    assert lib.aud_plan_create(ctx.handle, C.byref(d), None, None, None, C.byref(h)) == capi.AUD_EINVAL      # no tables
    dftp = capi.DftParams(1, -100.0, 1.0, 0.0, 1.0)
    fb = capi.MelFBank(32, 0.0, 8000.0, 0.0, -10.0, 0, -6.0, 4.0, 0.0)
    for kw in (dict(compute_dtype=7), dict(mfcc_coefs=33), dict(mfcc_coefs=-1)):
        with pytest.raises(capi.AuditoryError):
            runtime.Plan(ctx, 400, 160, 14, 2, dftp, fb, oc.bins, oc.filt, **kw)
    with pytest.raises(capi.AuditoryError):                                                 # bad step
        runtime.Plan(ctx, 400, 0, 14, 2, dftp, fb, oc.bins, oc.filt)
    assert lib.aud_plan_destroy(None) == capi.AUD_EINVAL
    assert lib.aud_plan_set_option(None, b"kernel", 1) == capi.AUD_EINVAL
    assert lib.aud_last_error(None) == b"null context"
    assert lib.aud_shutdown(None) == capi.AUD_EINVAL


def test_per_step_and_comm_errors(orc, emu):
    oc = W.OracleCfg(orc, "sndenv_16k_n400_nf32")
    plan = W.product_plan(oc)
    lib, h = plan.lib, plan.handle
    win, pw = np.zeros(400), np.zeros(201)
    seg = np.zeros((201, 14))
    assert lib.aud_dft_filter_host(h, 14, _vp(win), _vp(pw), _vp(pw), _vp(seg), _vp(seg)) == capi.AUD_EINVAL   # step >= T
    assert lib.aud_dft_filter_host(h, 0, None, _vp(pw), _vp(pw), _vp(seg), _vp(seg)) == capi.AUD_EINVAL
    assert lib.aud_mel_filter_dft_host(h, -1, _vp(pw), _vp(seg), None) == capi.AUD_EINVAL
    sig = np.arange(10.0)
    assert lib.aud_snd_to_window(_vp(sig), 10, 8, 4, _vp(win)) == capi.AUD_ESHORT          # end beyond signal
    assert lib.aud_snd_to_window(_vp(sig), 10, -2, 4, _vp(win)) == 0 and list(win[:4]) == [0, 0, 0, 1]
    # the all-gather needs a communicator
    assert lib.aud_allgather_dev(plan.ctx.handle, _vp(win), _vp(win), 4, None) == capi.AUD_ERCCL
    assert lib.aud_comm_init(plan.ctx.handle, 2, 5, _vp(win)) == capi.AUD_EINVAL            # rank >= n_ranks
    assert lib.aud_comm_destroy(plan.ctx.handle) == 0
    plan.close()


def test_resident_signal_argument_errors(orc, emu):
    """aud_signal_* / the _sig entry points: misuse is a status code, nothing is written"""
    oc = W.OracleCfg(orc, "sndenv_16k_n400_nf32")
    plan = W.product_plan(oc)
    lib, h, ctx = plan.lib, plan.handle, plan.ctx.handle
    sig = np.zeros(4000)
    hs = C.c_void_p()
    assert lib.aud_signal_upload(ctx, _vp(sig), 7, 4000, C.byref(hs)) == capi.AUD_EINVAL          # bad sample type
    assert lib.aud_signal_upload(ctx, None, capi.AUD_F64, 4000, C.byref(hs)) == capi.AUD_EINVAL  # null samples
    assert lib.aud_signal_upload(ctx, _vp(sig), capi.AUD_F64, -1, C.byref(hs)) == capi.AUD_EINVAL
    assert lib.aud_signal_upload(None, _vp(sig), capi.AUD_F64, 4000, C.byref(hs)) == capi.AUD_EINVAL
    assert not hs.value
    assert lib.aud_signal_upload(ctx, _vp(sig), capi.AUD_F64, 4000, C.byref(hs)) == capi.AUD_OK and hs.value
    assert lib.aud_signal_len(hs) == 4000 and lib.aud_signal_len(None) == -1
    items = runtime.make_items([0], [4000], [0])
    mel = np.full((1, 32, 14), 7.0)
    assert lib.aud_melspec_batch_sig(h, None, _vp(items), 1, _vp(mel), None, None) == capi.AUD_EINVAL      # null signal
    assert lib.aud_melspec_batch_sig(h, hs, None, 1, _vp(mel), None, None) == capi.AUD_EINVAL
    assert lib.aud_melspec_batch_sig(h, hs, _vp(items), 1, None, None, None) == capi.AUD_EINVAL
    bad = runtime.make_items([1], [4000], [0])                                                            # one sample past its end
    assert lib.aud_melspec_batch_sig(h, hs, _vp(bad), 1, _vp(mel), None, None) == capi.AUD_EINVAL
    assert b"outside" in lib.aud_last_error(ctx)
    # the MFCC form on a plan created without mfcc_coefs
    assert lib.aud_melspec_mfcc_batch_sig(h, hs, _vp(items), 1, _vp(mel), None, None, _vp(mel), None, None, None) == capi.AUD_EINVAL
    assert (mel == 7.0).all()
    assert lib.aud_melspec_batch_sig(h, hs, _vp(items), 0, None, None, None) == capi.AUD_OK               # empty batch
    assert lib.aud_signal_destroy(hs) == capi.AUD_OK and lib.aud_signal_destroy(None) == capi.AUD_EINVAL
    # an empty signal is a valid (useless) one
    assert lib.aud_signal_upload(ctx, None, capi.AUD_I16, 0, C.byref(hs)) == capi.AUD_OK
    assert lib.aud_signal_len(hs) == 0 and lib.aud_signal_destroy(hs) == capi.AUD_OK
    plan.close()


def test_signal_sync_and_live_argument_errors(orc, emu):
    """aud_signal_sync / aud_melspec_batch_live / aud_melspec_mfcc_batch_live: misuse is a status code, nothing is written;
    the handle lives across type and length changes; a destroyed or foreign handle is refused"""
    oc = W.OracleCfg(orc, "sndenv_16k_n400_nf32")
    plan = W.product_plan(oc)
    lib, h, ctx = plan.lib, plan.handle, plan.ctx.handle
    sig = np.linspace(-0.5, 0.5, 4000)
    hs, up = C.c_void_p(), C.c_int64(-1)
    assert lib.aud_signal_sync(ctx, None, _vp(sig), capi.AUD_F64, 4000, C.byref(up)) == capi.AUD_EINVAL      # no handle slot
    assert lib.aud_signal_sync(None, C.byref(hs), _vp(sig), capi.AUD_F64, 4000, C.byref(up)) == capi.AUD_EINVAL
    assert lib.aud_signal_sync(ctx, C.byref(hs), _vp(sig), 9, 4000, C.byref(up)) == capi.AUD_EINVAL          # bad sample type
    assert lib.aud_signal_sync(ctx, C.byref(hs), None, capi.AUD_F64, 4000, C.byref(up)) == capi.AUD_EINVAL   # null samples
    assert lib.aud_signal_sync(ctx, C.byref(hs), _vp(sig), capi.AUD_F64, -3, C.byref(up)) == capi.AUD_EINVAL
    assert not hs.value and up.value == 0
    assert lib.aud_signal_sync(ctx, C.byref(hs), _vp(sig), capi.AUD_F64, 4000, C.byref(up)) == capi.AUD_OK and hs.value
    assert up.value == 32000 and lib.aud_signal_len(hs) == 4000
    assert lib.aud_signal_sync(ctx, C.byref(hs), _vp(sig), capi.AUD_F64, 4000, None) == capi.AUD_OK          # (the count is optional)
    first = hs.value
    s32 = sig.astype(np.float32)                                                                             # another type, same handle
    assert lib.aud_signal_sync(ctx, C.byref(hs), _vp(s32), capi.AUD_F32, 4000, C.byref(up)) == capi.AUD_OK
    assert hs.value == first and up.value == 16000
    assert lib.aud_signal_sync(ctx, C.byref(hs), _vp(sig), capi.AUD_F64, 3000, C.byref(up)) == capi.AUD_OK and up.value == 24000
    assert lib.aud_signal_len(hs) == 3000
    items = runtime.make_items([0], [3000], [0])
    mel = np.full((1, 32, 14), 7.0)
    assert lib.aud_melspec_batch_live(h, None, _vp(sig), 3000, _vp(items), 1, _vp(mel), None, None, None) == capi.AUD_EINVAL
    assert lib.aud_melspec_batch_live(h, C.byref(hs), None, 3000, _vp(items), 1, _vp(mel), None, None, None) == capi.AUD_EINVAL
    assert lib.aud_melspec_batch_live(h, C.byref(hs), _vp(sig), 3000, None, 1, _vp(mel), None, None, None) == capi.AUD_EINVAL
    assert lib.aud_melspec_batch_live(h, C.byref(hs), _vp(sig), 3000, _vp(items), 1, None, None, None, None) == capi.AUD_EINVAL
    bad = runtime.make_items([1], [3000], [0])                                                               # one sample past the end
    assert lib.aud_melspec_batch_live(h, C.byref(hs), _vp(sig), 3000, _vp(bad), 1, _vp(mel), None, None, C.byref(up)) == capi.AUD_EINVAL
    assert b"outside" in lib.aud_last_error(ctx) and up.value == 0
    # the MFCC form on a plan created without mfcc_coefs
    assert lib.aud_melspec_mfcc_batch_live(h, C.byref(hs), _vp(sig), 3000, _vp(items), 1, _vp(mel), None, None, _vp(mel), None, None,
                                           None, None) == capi.AUD_EINVAL
    assert (mel == 7.0).all()
    assert lib.aud_melspec_batch_live(h, C.byref(hs), _vp(sig), 3000, _vp(items), 0, None, None, None, None) == capi.AUD_OK   # empty batch
    assert lib.aud_melspec_batch_live(h, C.byref(hs), _vp(sig), 3000, _vp(items), 1, _vp(mel), None, None, C.byref(up)) == capi.AUD_OK
    assert up.value == 0 and not (mel == 7.0).any()                                                          # current: nothing moved
    assert lib.aud_signal_destroy(hs) == capi.AUD_OK
    dead = C.c_void_p(first)                                                                                 # a destroyed handle is refused
    assert lib.aud_signal_sync(ctx, C.byref(dead), _vp(sig), capi.AUD_F64, 3000, None) == capi.AUD_EINVAL
    assert lib.aud_melspec_batch_live(h, C.byref(dead), _vp(sig), 3000, _vp(items), 1, _vp(mel), None, None, None) == capi.AUD_EINVAL
    # a fresh slot works again, from a live call alone
    hs = C.c_void_p()
    assert lib.aud_melspec_batch_live(h, C.byref(hs), _vp(sig), 3000, _vp(items), 1, _vp(mel), None, None, C.byref(up)) == capi.AUD_OK
    assert hs.value and up.value == 24000 and lib.aud_signal_destroy(hs) == capi.AUD_OK
    plan.close()


def test_process_batch_rejects_before_writing(orc, emu):
    """aud_process_batch_dev with pools that reach past the mel matrix (the Go code would panic in Convolve):
    AUD_EINVAL and NOTHING written -- the mel tensor included -- on the two-launch path and on the fused one"""
    import parity_cases as PC
    oc = W.OracleCfg(orc, "cfg2_16k_n400_nf40")
    L = oc.full_len()
    plan = W.product_plan(oc, capi.AUD_F64, PC.GABOR_DEFAULT)
    sig = np.zeros(L, np.float32)
    items = runtime.make_items([0], [L], [0])
    try:
        for ik in (-1, 1):
            plan.set_option("item_kernel", ik)
            mel = np.full((1, oc.nf, oc.T), 7.0, np.float32)
            gab = np.full((1, 12, 40, 2, 8), 7.0, np.float32)
            rc = plan.lib.aud_process_batch_dev(plan.handle, _vp(sig), capi.AUD_F32, _vp(items), 1, _vp(mel), 12, 40, _vp(gab), None)
            assert rc == capi.AUD_EINVAL and b"reach past" in plan.lib.aud_last_error(plan.ctx.handle)
            assert (mel == 7.0).all() and (gab == 7.0).all()
    finally:
        plan.close()
