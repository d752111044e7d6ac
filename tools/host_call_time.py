#!/usr/bin/env python3
"""Run ON THE GPU BOX: what the host-facing entry points cost end to end for 256 utterances of 1 s (the PCIe-inclusive rates
DESIGN.md 6 and BASELINE.md quote; never bench.py's `value`):
  host      aud_melspec_batch_host: the float64 Signal tensor (33.9 MB, pageable) copied in on every call, kernel, float32
            results back through pinned staging, widened into the caller's float64 tensors
  sig_f64   aud_melspec_batch_sig on a signal uploaded once (aud_signal_upload): items up, results back
  sig_i16   the same on int16 PCM uploaded once (8.5 MB, normalised on the device)
  *_reuse   ... into a caller-owned result tensor that is reused (no fresh pages per call)
  *_pinned  ... into a result tensor from aud_host_alloc: the device widens and writes it (no staging copy, no CPU pass)
  *_shared  ... into a mapping of POSIX shared memory registered with aud_host_register (what several processes, one per GPU,
            would write their shards of one host tensor into): the device writes it like the pinned one
  upload_*  the one-time uploads themselves
and, round 6, what the DEFAULT of the SndEnv mirrors costs per sound -- the resident copy validated exactly on every call
(aud_signal_sync: memcmp against a host shadow) -- one ProcessSegment call on one 3 s sound (SndEnv defaults: 100 ms segments):
  sound_snapshot   the _sig call alone on a snapshot (the opt-in)
  sound_r5_default round 5's default: the 64-sample fingerprint in Python, then the call on the snapshot
  sound_live       aud_melspec_batch_live: ONE call that compares the 4 KB blocks the segment's frames read with the resident
                   copy's shadow, uploads what differs, and runs -- the mirrors' default
  sound_live_edit  ... with one sample of the segment edited before every call (one block goes up)
  sound_sync       aud_signal_sync over the WHOLE unchanged tensor + the _sig call (two calls; what a caller gets who wants
                   every edit uploaded at once)
  sound_sync_edit  ... with one sample edited before every call (one 4 KB block goes up)
  sound_per_call   the tensor copied in on every call
  sync_8MB         aud_signal_sync alone on an unchanged tensor of AUD_RESIDENT_AUTO_BYTES (the largest the default validates)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as W  # noqa: E402
from auditory_amd import capi, runtime, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

oc = W.OracleCfg(orc, "cfg2_16k_n400_nf40")
L = oc.full_len()
n = 256
sig, pcm = synth.batch(3, n, 16000, oc.sr, row_len=L)
plan = W.product_plan(oc, capi.AUD_F64)
items = runtime.make_items(np.arange(n) * L, [L] * n, [0] * n)
flat = sig.ravel()


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps


res = {}
res["host"] = timed(lambda: plan.melspec_host(flat, items))
t0 = time.perf_counter(); s64 = runtime.Signal(plan.ctx, flat); res["upload_f64"] = time.perf_counter() - t0
t0 = time.perf_counter(); s16 = runtime.Signal(plan.ctx, pcm.ravel()); res["upload_i16"] = time.perf_counter() - t0
res["sig_f64"] = timed(lambda: plan.melspec_sig(s64, items))
res["sig_i16"] = timed(lambda: plan.melspec_sig(s16, items))
pin_mel = plan.ctx.pinned_empty((n, 40, 104))
res["sig_f64_pinned"] = timed(lambda: plan.melspec_sig(s64, items, out=(pin_mel, None, None)))
res["sig_i16_pinned"] = timed(lambda: plan.melspec_sig(s16, items, out=(pin_mel, None, None)))
import tempfile  # noqa: E402
fd, shm_path = tempfile.mkstemp(prefix="auditory_hip_time_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
os.close(fd)
shared = np.memmap(shm_path, dtype=np.float64, mode="w+", shape=(n, 40, 104))
shared[...] = 0.0
plan.ctx.register_host(shared)
res["sig_f64_shared"] = timed(lambda: plan.melspec_sig(s64, items, out=(shared, None, None)))
keep = np.zeros((n, 40, 104))
res["sig_f64_reuse"] = timed(lambda: plan.melspec_sig(s64, items, out=(keep, None, None)))
a, b = plan.melspec_host(flat, items)[0], plan.melspec_sig(s64, items)[0]
assert np.array_equal(a, b)
print("256 utterances of 1 s, float64 plan (w20x10), float64 mel %.1f MB out" % (n * 40 * 104 * 8 / 1e6))
assert np.array_equal(plan.melspec_sig(s64, items, out=(pin_mel, None, None))[0], b)
assert np.array_equal(np.asarray(shared), b)
plan.ctx.unregister_host(shared)
del shared
os.unlink(shm_path)
for k in ("host", "sig_f64", "sig_i16", "sig_f64_reuse", "sig_f64_pinned", "sig_i16_pinned", "sig_f64_shared"):
    print("  %-15s %.3f ms per call = %.0f audio-s/s" % (k, res[k] * 1e3, n / res[k]))
print("  one-time uploads: float64 signal (%.1f MB) %.2f ms, int16 PCM (%.1f MB) %.2f ms"
      % (flat.nbytes / 1e6, res["upload_f64"] * 1e3, pcm.nbytes / 1e6, res["upload_i16"] * 1e3))


# ---- per sound: the SndEnv default (one segment per call on a 3 s sound)
oc1 = W.OracleCfg(orc, "sndenv_16k_n400_nf32")
plan1 = W.product_plan(oc1, capi.AUD_F64)
snd = np.ascontiguousarray(synth.batch(5, 1, 48000, 16000)[0][0])
one = runtime.make_items([0], [len(snd)], [16000])
snap, live = runtime.Signal(plan1.ctx, snd), runtime.Signal(plan1.ctx)
live.sync(snd)
out1 = (np.zeros((1, plan1.nf, plan1.T)), np.zeros((1, plan1.H, plan1.T)), np.zeros((1, plan1.H, plan1.T)))
r1 = {}
r1["sound_snapshot"] = timed(lambda: plan1.melspec_sig(snap, one, out=out1), reps=200, warm=20)


def round5_default():   # what the Python mirror did per call until round 5: a fingerprint of <= 64 probed samples, then the call
    n_ = len(snd)
    probe = np.ascontiguousarray(snd[::max(1, n_ // 61)][:63]).tobytes() + np.asarray(snd[-1:]).tobytes()
    _ = (snd.__array_interface__["data"][0], n_, snd.dtype.str, snd.strides, hash(probe))
    plan1.melspec_sig(snap, one, out=out1)


r1["sound_r5_default"] = timed(round5_default, reps=200, warm=20)


def synced():
    live.sync(snd)
    plan1.melspec_sig(live, one, out=out1)


r1["sound_sync"] = timed(synced, reps=200, warm=20)
assert live.uploaded_bytes == 0
k = [0]


def edited():
    k[0] += 1
    snd[17000 + k[0]] += 1e-3
    synced()


r1["sound_sync_edit"] = timed(edited, reps=200, warm=20)
assert live.uploaded_bytes in (4096, 8192)
ref1 = plan1.melspec_host(snd, one, True, True)
plan1.melspec_sig(live, one, out=out1)
assert all(np.array_equal(a, b) for a, b in zip(ref1, out1))       # the edited tensor's features, without any announcement
r1["sound_per_call"] = timed(lambda: plan1.melspec_host(snd, one, True, True), reps=200, warm=20)
# the default of the mirrors since the second half of round 6: ONE call that compares only the blocks the segment's frames read
lv = runtime.Signal(plan1.ctx)
plan1.melspec_live(lv, snd, one, out=out1)
r1["sound_live"] = timed(lambda: plan1.melspec_live(lv, snd, one, out=out1), reps=200, warm=20)
assert lv.uploaded_bytes == 0


def live_edited():
    k[0] += 1
    snd[17000 + k[0]] += 1e-3
    plan1.melspec_live(lv, snd, one, out=out1)


r1["sound_live_edit"] = timed(live_edited, reps=200, warm=20)
assert lv.uploaded_bytes in (4096, 8192)
ref1 = plan1.melspec_host(snd, one, True, True)
assert all(np.array_equal(a, b) for a, b in zip(ref1, out1))
big = np.zeros(capi.AUD_RESIDENT_AUTO_BYTES // 8)
big[::7] = 0.25
bsig = runtime.Signal(plan1.ctx)
bsig.sync(big)
r1["sync_8MB"] = timed(lambda: bsig.sync(big), reps=50, warm=5)
# the same three routes at the C ABI with the ctypes arguments built ONCE (what a compiled host -- Go through cgo, C++ -- pays:
# the rows above include 1-2 us of Python marshalling per pointer argument, and the live call has three more of them)
import ctypes as C  # noqa: E402
vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
lib1, up1 = plan1.lib, C.c_int64(0)
a_sig = (plan1.handle, snap.handle, vp(one), 1, vp(out1[0]), vp(out1[1]), vp(out1[2]))
a_live = (plan1.handle, C.byref(lv.handle), vp(snd), snd.size, vp(one), 1, vp(out1[0]), vp(out1[1]), vp(out1[2]), C.byref(up1))
a_sync = (plan1.ctx.handle, C.byref(live.handle), vp(snd), capi.AUD_F64, snd.size, C.byref(up1))
a_sig2 = (plan1.handle, live.handle, vp(one), 1, vp(out1[0]), vp(out1[1]), vp(out1[2]))
assert lib1.aud_melspec_batch_live(*a_live) == 0 and lib1.aud_signal_sync(*a_sync) == 0
r1["abi_snapshot"] = timed(lambda: lib1.aud_melspec_batch_sig(*a_sig), reps=400, warm=40)
r1["abi_live"] = timed(lambda: lib1.aud_melspec_batch_live(*a_live), reps=400, warm=40)
r1["abi_sync_whole"] = timed(lambda: (lib1.aud_signal_sync(*a_sync), lib1.aud_melspec_batch_sig(*a_sig2)), reps=400, warm=40)
r1["abi_snapshot_again"] = timed(lambda: lib1.aud_melspec_batch_sig(*a_sig), reps=400, warm=40)
print("one 3 s sound (384 KB of float64), one 100 ms segment per call (mel + PowerSegment + LogPowerSegment out):")
for k_ in ("sound_snapshot", "sound_r5_default", "sound_live", "sound_live_edit", "sound_sync", "sound_sync_edit", "sound_per_call", "sync_8MB"):
    print("  %-16s %.1f us per call" % (k_, r1[k_] * 1e6))
print("  exact residency (sound_live, the mirrors' default): %+.1f %% against the bare snapshot call, %+.1f %% against round 5's default "
      "(sampled fingerprint + call); comparing the WHOLE tensor per call (sound_sync): %+.1f %% / %+.1f %%"
      % (100.0 * (r1["sound_live"] / r1["sound_snapshot"] - 1.0), 100.0 * (r1["sound_live"] / r1["sound_r5_default"] - 1.0),
         100.0 * (r1["sound_sync"] / r1["sound_snapshot"] - 1.0), 100.0 * (r1["sound_sync"] / r1["sound_r5_default"] - 1.0)))
print("  at the C ABI, arguments prebuilt: snapshot %.1f / %.1f us, live %.1f us (%+.1f %%), sync of the whole tensor + call %.1f us (%+.1f %%)"
      % (r1["abi_snapshot"] * 1e6, r1["abi_snapshot_again"] * 1e6, r1["abi_live"] * 1e6,
         100.0 * (r1["abi_live"] / min(r1["abi_snapshot"], r1["abi_snapshot_again"]) - 1.0), r1["abi_sync_whole"] * 1e6,
         100.0 * (r1["abi_sync_whole"] / min(r1["abi_snapshot"], r1["abi_snapshot_again"]) - 1.0)))
