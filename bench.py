#!/usr/bin/env python3
"""bench.py -- audio-seconds/sec of the frame -> FFT -> power -> mel (-> gabor) hot path on MI355X.

One GPU (`python bench.py`): the headline is the metric's own parameter set on BASELINE.json configs[1]'s batch: 256
synthetic 16 kHz mono utterances of 1 s per step, WinMs 25 (N = 400, no taper), StepMs 10 (S = 160), one segment per
utterance with BorderSteps 2 (T = 104 frames), 40 mel filters 0-8000 Hz, mel output only, float64 arithmetic (the
reference's).  A step = one launch of the fused kernel over one resident batch; consecutive steps walk a ring of resident
batches larger than the 256 MB Infinity Cache, so the input really comes from HBM.  The other BASELINE configurations are
measured in the same run and nested under `also`, each with its own strict parity object: configs[1] as worded (WinMs 32,
N = 512), configs[2] on one GPU (4096 utterances per step), configs[3] (+ agabor.Convolve) and configs[4] (44.1 kHz 5 s
streams, N = 2048, 128 mel, 1.13 GB per batch); the float32 instantiations are under `modes`.

Several GPUs (`python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P
bench.py --gpus G --steps K --warmup W`, one rank per GPU): `value` is BASELINE configs[2] AS STATED -- 4096 utterances per
step in total, a contiguous shard of 4096 / G per rank (auditory_amd.batch.shard_range), and the path's one collective,
the RCCL all-gather that reassembles the [4096, 40, 104] float32 feature tensor on every rank, issued on a second stream
and overlapped with the next step's kernel; strong scaling.  Rank 0 checks the GATHERED tensor against the oracle.  The
collective-free sharded step (256 utterances per rank) is reported beside it as `no_collective`.

Every timed region is captured into one hipGraph (a ~12 us kernel is otherwise bound by the Python launch path) --
the K steps asked for, repeated until the graph holds at least --graph-steps steps, so that the fork / join of the
streams at the ends of a replay is not what is timed -- and replayed back to back until at least --min-seconds of device
time have passed; `steps` is the number of steps actually timed, bracketed by a barrier + synchronize on both sides,
max over ranks.  Inside a graph consecutive steps alternate between two streams (independent batches with their own
output buffers): a launch of 256 utterances is a burst of little more than one round of resident waves whose load phase
and tail leave the chip half idle, and overlapping step i+1's start with step i's tail is what a double-buffered pipeline
does.  `roofline` is priced on the kernel ALONE (a one-stream region of the same run).

Every mode carries a `parity` object: timed outputs checked against the oracle under |d| <= 1e-5 max(1, |ref|) on every
element; a headline mode with an element past it makes the run exit non-zero without a JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import memguard  # noqa: E402  resident-memory ceiling (tools/memguard.py)

memguard.install()

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VECTOR_PEAK_TF = {"f64": 78.6, "f32": 157.3}  # MI355X_MICROARCH.md: vector (non-matrix) peaks of the arithmetic types
METRIC = "audio-seconds/sec (16 kHz, 25 ms/10 ms, 40 mel) at 1/2/4/8 MI355X"
TOL = 1e-5              # BASELINE.json north_star: 1e-5 relative on the float32 mel / gabor tensors

# name: sample rate, WinMs, StepMs, SegmentMs (= StrideMs), BorderSteps, mel filters, LoHz, HiHz, seconds of audio per stream
WORKLOADS = {
    "n400": (16000, 25.0, 10.0, 1000.0, 2, 40, 0.0, 8000.0, 1.0),     # the metric's parameters
    "n512": (16000, 32.0, 10.0, 1000.0, 2, 40, 0.0, 8000.0, 1.0),     # BASELINE configs[1] as worded ("512-pt FFT")
    "cfg5": (44100, 46.44, 10.0, 5000.0, 2, 128, 0.0, 22050.0, 5.0),  # BASELINE configs[4]
    # BASELINE configs[0]'s parameter set (processspeech defaults on the shipped 44.1 kHz WAVs: N = 1103, prime; 100 ms
    # segments of 14 steps, 32 mel): every work item is one segment -- the generic any-N kernel, a stated non-headline row
    "cfg1": (44100, 25.0, 10.0, 100.0, 2, 32, 0.0, 8000.0, 0.1),
    # processspeech's parameters on 48 kHz audio (N = 1200 = 2^4 3 5^2): the any-N kernel's in-place route for smooth lengths
    "rate48k": (48000, 25.0, 10.0, 100.0, 2, 32, 0.0, 8000.0, 0.1),
}
GABOR_SPECS = [dict(WaveLen=2.0, Orientation=o, SigmaWidth=0.5, SigmaLength=0.5, PhaseOffset=ph, CircleEdge=True)
               for o in (0, 45, 90, 135) for ph in (0, 1.5708)]        # processspeech.go:236-252
GABOR_POOLS = (11, 32)                                                 # [PoolsY, PoolsX] of the [.., 2, 8] output


class Workload:
    """Geometry and tables of one parameter set, from the PRODUCT's own host setup (aud_sound_params_derive,
    mel.Params.InitFilters, agabor.ToTensor); the oracle is not involved."""

    def __init__(self, name):
        from auditory_amd import capi, mel
        self.name = name
        self.sr, self.win_ms, self.step_ms, self.seg_ms, self.border, self.nf, self.lo, self.hi, self.dur_s = WORKLOADS[name]
        lib = capi.load()
        sp = capi.SoundParams()
        lib.aud_sound_params_defaults(sp)
        sp.win_ms, sp.step_ms, sp.segment_ms, sp.stride_ms, sp.border_steps = (self.win_ms, self.step_ms, self.seg_ms,
                                                                                 self.seg_ms, self.border)
        if lib.aud_sound_params_derive(sp, self.sr) != capi.AUD_OK:
            raise RuntimeError("aud_sound_params_derive failed")
        self.N, self.S, self.T = sp.win_samples, sp.step_samples, sp.segment_steps
        self.H = self.N // 2 + 1
        self.mp = mel.Params()
        self.mp.Defaults()
        self.mp.FBank.NFilters, self.mp.FBank.LoHz, self.mp.FBank.HiHz = self.nf, self.lo, self.hi
        self.filt = self.mp.InitFilters(self.N, self.sr)
        self.dur = int(round(self.dur_s * self.sr))                    # samples of audio per stream
        need = self.S * (self.T - 1 - self.border) + self.N            # every frame of the segment in bounds
        self.L = (max(need, self.dur) + 63) // 64 * 64                 # row pitch: zero tail, 64-sample multiple

    def plan(self, compute, device, gabor=False, mfcc=0):
        from auditory_amd import agabor, capi, runtime
        dftp = capi.DftParams()
        capi.load().aud_dft_defaults(dftp)
        gset = gk = None
        if gabor:
            fs = agabor.FilterSet()
            fs.SizeX, fs.SizeY, fs.StrideX, fs.StrideY, fs.Gain = 9, 9, 3, 3, 2.0
            agabor.ToTensor([agabor.Filter(**s) for s in GABOR_SPECS], fs)
            gset, gk = fs.to_c(), fs.Filters
        cdt = capi.AUD_F64 if compute == "f64" else capi.AUD_FAST_F32
        return runtime.Plan(runtime.get_ctx(device), self.N, self.S, self.T, self.border, dftp, self.mp.FBank.to_c(),
                            self.mp.BinPts, self.filt, gset, gk, cdt, mfcc_coefs=mfcc)

    def describe(self, B):
        return ("%d synthetic %g kHz mono streams of %g s per GPU and step, WinMs %g (N = %d), StepMs %g (S = %d), "
                "T = %d frames, %d mel" % (B, self.sr / 1e3, self.dur_s, self.win_ms, self.N, self.step_ms,
                                                     self.S, self.T, self.nf))


class OracleSide:
    """The checker: oracle parameter blocks for a workload (tests/, smoke() and this file's parity / cpu_baseline legs
    are the only users of oracle/)."""

    def __init__(self, wl):
        from oracle import oracle as orc
        self.orc = orc
        self.sp = orc.sound_params(wl.win_ms, wl.step_ms, wl.seg_ms, wl.seg_ms, wl.border, wl.sr)
        self.d = orc.dft_defaults()
        self.m = orc.mel_defaults()
        self.m.n_filters, self.m.lo_hz, self.m.hi_hz = wl.nf, wl.lo, wl.hi
        rc, self.bins, _, self.filt = orc.mel_init_filters(self.m, self.sp.win_samples, wl.sr)
        assert rc == 0
        assert (self.sp.win_samples, self.sp.step_samples, self.sp.segment_steps) == (wl.N, wl.S, wl.T), \
            "product and oracle derive different geometry"

    def mel(self, rows64, gabor=False):
        """[n, L] float64 rows -> [n, nf, T] oracle mel (and the [n, 11, 32, 2, 8] gabor tensor of the oracle's mel)"""
        n, L = rows64.shape
        g = None
        if gabor:
            g = dict(k=self.orc.gabor_to_tensor([dict(wave_len=s["WaveLen"], orientation=s["Orientation"],
                                                      sigma_width=s["SigmaWidth"], sigma_length=s["SigmaLength"],
                                                      phase_offset=s["PhaseOffset"], circle_edge=1) for s in GABOR_SPECS], 9, 9),
                     stride_x=3, stride_y=3, gain=2.0, py=GABOR_POOLS[0], px=GABOR_POOLS[1])
        rc, mel, gab = self.orc.process_batch(self.sp, self.d, self.m, self.bins, self.filt, rows64.reshape(-1),
                                              np.arange(n) * L, np.full(n, L), np.zeros(n), gabor=g)
        assert rc == 0
        return (mel, gab) if gabor else mel


def cpu_baseline(wl, pcm, target_s=12.0, max_threads=16, chunk=8, all_seconds=5.0):
    """The oracle (C float64 restatement of the reference, FFT plan cached per segment) timed on this box's host cores,
    `chunk` utterances per call, one call per thread at a time (ctypes releases the GIL).

    Memory is bounded by construction: the sample is at most 256 utterances converted once to float64 (a few tens of MB),
    every call gets a VIEW of it, the thread count is capped at the GPU box's CPU share (16 per GPU) and the number of
    calls per thread at 2000.  (Round 1 copied ~1 GB per thread and took two GPU boxes down.)"""
    from concurrent.futures import ThreadPoolExecutor
    osd = OracleSide(wl)
    orc = osd.orc
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, max_threads))
    rows = pcm[:256].astype(np.float64) / 32767.0               # sound.go:138
    n_rows, L = rows.shape
    chunk = min(chunk, n_rows)
    flat = rows.reshape(-1)

    def run_chunk(first, faithful=False):
        first = first % (n_rows - chunk + 1)
        view = flat[first * L:(first + chunk) * L]                # contiguous view, no copy
        rc, _, _ = orc.process_batch(osd.sp, osd.d, osd.m, osd.bins, osd.filt, view, np.arange(chunk) * L,
                                     np.full(chunk, L), np.zeros(chunk), faithful=faithful)
        assert rc == 0
        return chunk

    t0 = time.perf_counter()
    run_chunk(0)
    per_utt = (time.perf_counter() - t0) / chunk
    t0 = time.perf_counter()
    run_chunk(0, faithful=True)
    per_utt_faithful = (time.perf_counter() - t0) / chunk
    calls = int(min(2000, max(1, target_s / (per_utt * chunk))))

    def worker(t):
        return sum(run_chunk(t * 131 + c * chunk) for c in range(calls))

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        n = sum(ex.map(worker, range(cores)))
    dt = time.perf_counter() - t0

    # SURVEY 8d's "all host cores" flavour: one thread per CPU of the affinity mask, every thread running chunks until a
    # DEADLINE (bounded by wall time, not by a call count: where a CPU quota holds the process to fewer cores than the mask
    # shows, 256 threads x a fixed amount of work would run for minutes)
    all_cores = None
    if all_seconds > 0 and avail > cores:
        def until(t):
            end, c, done = t_start[0] + all_seconds, 0, 0
            while time.perf_counter() < end:
                done += run_chunk(t * 131 + c * chunk)
                c += 1
            return done
        t_start = [time.perf_counter()]
        with ThreadPoolExecutor(avail) as ex:
            n_all = sum(ex.map(until, range(avail)))
        dt_all = time.perf_counter() - t_start[0]
        quota = None
        try:
            q = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q[0] == "max" else round(float(q[0]) / float(q[1]), 2)
        except (OSError, ValueError, IndexError):
            pass
        all_cores = {"value": round(n_all * wl.dur_s / dt_all, 2), "threads": avail, "seconds": round(dt_all, 2),
                     "cgroup_cpu_quota": quota,
                     "note": "one thread per CPU of the affinity mask, each running %d-utterance calls until a %g s deadline; "
                             "cgroup_cpu_quota = CPUs the container may use at once (null: no quota)" % (chunk, all_seconds)}
    return {"value": round(n * wl.dur_s / dt, 2), "unit": "audio-seconds/sec", "cores": cores, "kind": "port", "all_cores": all_cores,
            "affinity_cores": avail, "box_cores_online": os.cpu_count(),   # threads used = min(affinity, 16): SURVEY 8d asks for the box's count
            "sample": "%d utterance passes over the first %d utterances of the bench ring (%s), oracle/auditory_oracle.c "
                      "float64, %d threads x %d calls x %d utterances, FFT plan cached per segment"
                      % (n, n_rows, wl.name, cores, calls, chunk),
            "one_thread_cached": round(wl.dur_s / per_utt, 2),
            "one_thread_plan_per_frame": round(wl.dur_s / per_utt_faithful, 2)}


_real_cpu_baseline = cpu_baseline


def frame_flops(wl):
    """SURVEY 8d's algorithmic flops of one frame hop: real-to-complex FFT 2.5 N log2 N + power 3 H + mel 2 x (sum of the
    triangles' widths) + one logarithm per filter"""
    widths = sum(int(wl.mp.BinPts[f + 2]) - int(wl.mp.BinPts[f]) + 1 for f in range(wl.nf))
    return 2.5 * wl.N * math.log2(wl.N) + 3.0 * wl.H + 2.0 * widths + wl.nf


def measured_stream_read(torch, dev, gib=2.0, reps=10):
    """SURVEY 8d's denominator: GB/s of a plain float32 read kernel (tools/ubench/stream_read.hip, built by
    auditory_amd.build.build_stream_read) over `gib` GiB -- eight times the 256 MB Infinity Cache -- in THIS process, on this
    box, timed between HIP events on the stream it is launched on.  None (with the reason) when the helper is missing."""
    import ctypes as C
    path = os.path.join(ROOT, "tools", "ubench", "libstream_read.so")
    if dev.type != "cuda":
        return None, "no GPU (CPU dry run)"
    if not os.path.exists(path):
        return None, "tools/ubench/libstream_read.so not built"
    lib = C.CDLL(path)
    lib.ubench_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ubench_stream_read.restype = C.c_int
    nbytes = int(gib * (1 << 30)) // 16 * 16
    buf = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
    sink = torch.zeros(256 * 32, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    best, out = None, {}
    for wgs in (1, 2, 4, 16):  # workgroups of 256 threads per CU: the best of four launch shapes is the roof
        gbps, ms = C.c_double(0.0), C.c_double(0.0)
        rc = lib.ubench_stream_read(buf.data_ptr(), nbytes, reps, wgs, torch.cuda.current_stream(dev).cuda_stream,
                                    sink.data_ptr(), C.byref(gbps), C.byref(ms))
        if rc != 0:
            return None, "ubench_stream_read rc %d" % rc
        out["%d_wgs_per_cu" % wgs] = round(gbps.value, 1)
        best = gbps.value if best is None else max(best, gbps.value)
    del buf, sink
    torch.cuda.empty_cache()
    return best, {"bytes_per_pass": nbytes, "passes": reps, "GBps_by_launch_shape": out,
                  "kernel": "k_stream_read (tools/ubench/stream_read.hip): 16-byte loads, 8 in flight per lane, nothing written"}


def strict_parity(got, ref):
    """north-star criterion on every element; returns the `parity` object"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    nan_ok = bool(np.array_equal(np.isnan(got), np.isnan(ref)))
    ok = ~np.isnan(ref)
    err = np.abs(got[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok]))
    return {"criterion": "|got - ref| <= 1e-5 * max(1, |ref|), every element", "elements": int(err.size),
            "max_scaled_err": float(err.max()) if err.size else 0.0, "n_past_1e-5": int((err > TOL).sum()),
            "p99.99": float(np.quantile(err, 0.9999)) if err.size else 0.0, "median": float(np.median(err)) if err.size else 0.0,
            "nan_pattern_equal": nan_ok, "pass": bool(nan_ok and (err.size == 0 or err.max() <= TOL))}


class Ring:
    """R resident batches of B streams each ([B, L] float32 or int16 on the device) + their outputs"""

    def __init__(self, torch, wl, B, R, rank, dev, sig_dtype, stereo=False, distinct=None):
        from auditory_amd import runtime, synth
        self.B, self.R, self.wl = B, R, wl
        assert not stereo or B % 2 == 0
        n_rows = R * B
        distinct = min(n_rows, distinct or n_rows)                    # long streams: `distinct` seeded streams, repeated
        base = np.zeros((distinct, wl.L), np.int16)                   # host copy (parity / cpu_baseline), 2 B per sample
        for i in range(distinct):
            base[i, :wl.dur] = synth.utterance_pcm(2, rank * n_rows + i, wl.dur, wl.sr)
        self.pcm = base if distinct == n_rows else base[np.arange(n_rows) % distinct]
        self.sig = []
        for r in range(R):
            blk = self.pcm[r * B:(r + 1) * B]
            host = blk if sig_dtype == "i16" else (blk.astype(np.float64) / 32767.0).astype(np.float32)
            if stereo:  # streams 2c and 2c + 1 are the channels of clip c: [B/2, L, 2] interleaved, as a WAV holds them
                host = host.reshape(B // 2, 2, wl.L).transpose(0, 2, 1)
            self.sig.append(torch.from_numpy(np.ascontiguousarray(host)).to(dev).view(-1))
        if stereo:  # two work items over one buffer: channel = sig_off parity, sig_stride 2 (aud_item)
            items = runtime.make_items((np.arange(B) // 2) * (2 * wl.L) + (np.arange(B) % 2), [wl.L] * B, [0] * B, sig_stride=2)
        else:
            items = runtime.make_items(np.arange(B) * wl.L, [wl.L] * B, [0] * B)
        raw = np.frombuffer(np.ascontiguousarray(items).tobytes(), np.uint8).copy()
        self.items = torch.from_numpy(raw).to(dev)
        self.mel = [torch.empty((B, wl.nf, wl.T), dtype=torch.float32, device=dev) for _ in range(R)]
        self.sample_bytes = 2 if sig_dtype == "i16" else 4

    def host_rows64(self, r, idx):
        """what the device saw for rows idx of ring buffer r, as float64 (float32 samples: the rounded values)"""
        blk = self.pcm[r * self.B + np.asarray(idx)].astype(np.float64) / 32767.0
        return blk if self.sample_bytes == 2 else blk.astype(np.float32).astype(np.float64)


class LineGuard:
    """Rank 0 of a multi-GPU run: a child process, forked BEFORE anything touches the GPU, that holds a copy of the JSON line
    while an optional measurement runs.  The parent sends the finished line down a pipe (`stash`), runs the measurement, and
    says `done` before it prints the final line itself; if the parent ends without saying so -- a GPU fault aborts the process,
    the launcher kills it because another rank died -- the child prints the stashed line (its stdout is the parent's).
    The child never imports or calls anything: it reads a pipe and writes one line."""

    def __init__(self):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            try:
                import signal
                for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):   # (a launcher that ends the ranks of a failed job
                    signal.signal(sig, signal.SIG_IGN)                       #  signals their whole group: outlive the parent)
                os.close(w)
                data = b""
                while True:
                    chunk = os.read(r, 1 << 16)
                    if not chunk:
                        break
                    data += chunk
                msgs = [m for m in data.split(b"\n") if m]
                if msgs and msgs[-1] != b"DONE":
                    os.write(1, msgs[-1] + b"\n")
            finally:
                os._exit(0)
        os.close(r)
        self.w, self.pid = w, pid

    def stash(self, line):
        os.write(self.w, json.dumps(line).encode() + b"\n")

    def done(self):
        os.write(self.w, b"DONE\n")
        os.close(self.w)
        os.waitpid(self.pid, 0)


def main():  # noqa: C901
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--workload", choices=["headline", "n512", "cfg4", "cfg5", "cfg1", "rate48k", "sndenv", "sndenv_cfg1"], default="headline",
                    help="headline: the judged line (N = 400; the other BASELINE configurations nested under `also`).  "
                         "Stand-alone lines for BASELINE.md's table: cfg4 = headline + agabor.Convolve (default FilterSet, "
                         "[11,32,2,8] pools); cfg5 = 44.1 kHz 5 s streams, N = 2048, 128 mel (use --batch 1280 for >= 1 GB "
                         "resident input); cfg1 = configs[0]'s parameters (N = 1103, prime); sndenv = the headline parameters "
                         "with everything the unmodified SndEnv.ProcessSegment loop produces (SURVEY 8 f-1 + f-2): mel + Power "
                         "+ LogPower tensors + the MFCC tail (13 coefficients, deltas, delta-deltas, Energy); sndenv_cfg1 = the same "
                         "on configs[0]'s parameters (N = 1103: what an unmodified SndEnv loop over the shipped 44.1 kHz WAVs runs)")
    ap.add_argument("--compute", choices=["f64", "f32"], default="f64", help="arithmetic of the headline mode")
    ap.add_argument("--sig-dtype", choices=["f32", "i16"], default="f32",
                    help="resident sample format: float32 (the metric's definition) or int16 PCM normalised on the device "
                         "(sound.go:116-141; half the input bytes)")
    ap.add_argument("--stereo", action="store_true",
                    help="the resident streams are the channels of interleaved stereo clips (BASELINE configs[4] as worded): "
                         "two strided work items per clip over one buffer, no de-interleaving copy")
    ap.add_argument("--ring-mb", type=float, default=320.0, help="resident input ring per GPU (> the 256 MB Infinity Cache)")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="minimum device time of the headline's timed regions")
    ap.add_argument("--also-seconds", type=float, default=0.15, help="the same for the modes under `also` / `modes`")
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph")
    ap.add_argument("--graph-steps", type=int, default=200,
                    help="steps one captured hipGraph holds: the requested --steps repeated until at least this many, so "
                         "that the fork/join of the streams at the two ends of a replay is not what is timed")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the steps are dealt over round-robin inside the graph (independent resident batches, "
                         "own output buffers): 1 = every step waits for the previous one; 2 (default) = consecutive launches "
                         "overlap their load burst with the previous launch's tail, as a double-buffered pipeline does.  Default "
                         "for cfg4 / sndenv: 4 -- a step is two dependent launches there.  `roofline` is always taken from a "
                         "1-stream region: one kernel alone on the chip, the duration rocprofv3 reports")
    ap.add_argument("--tail", choices=["fused", "parts"], default="fused",
                    help="--workload sndenv: aud_segment_batch_dev (default) or aud_melspec_batch_dev + aud_mfcc_batch_dev")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="aud_plan_set_option switches, e.g. kernel=1 (the generic kernel)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for CPU dry runs)")
    ap.add_argument("--gather-mode", choices=["rccl", "direct", "host"], default="rccl",
                    help="the multi-GPU region's collective: rccl = ncclAllGather (the reported default); direct = the library's "
                         "direct pattern (aud_gather_*: one device-to-device push per peer over its own xGMI link, SURVEY 5) -- the "
                         "fallback if RCCL picks a ring for these 8.5 MB slabs")
    ap.add_argument("--no-direct-alt", action="store_true",
                    help="N > 1: do not also time the step with the direct-pattern reassembly (side key `direct_gather`)")
    ap.add_argument("--dist-timeout", type=float, default=180.0,
                    help="seconds: torch.distributed's process-group timeout AND the bound on the first (eager) step with the "
                         "collective in it -- a rank that never arrives makes every other rank exit non-zero with the reason "
                         "instead of hanging in a captured collective")
    ap.add_argument("--dist-single", action="store_true",
                    help="validation only: run the multi-GPU code path (shard + RCCL all-gather inside the graph) on ONE rank")
    ap.add_argument("--only-headline", action="store_true", help="skip `modes` and `also`")
    ap.add_argument("--cfg3-total", type=int, default=4096, help="total utterances of the configs[2] region (CPU dry runs shrink it)")
    ap.add_argument("--cfg5-batch", type=int, default=1280, help="streams per step of the configs[4] region (1280 x 5 s = 1.13 GB)")
    ap.add_argument("--report-anyway", action="store_true",
                    help="stand-alone secondary rows only: print the line (with parity.pass = false) when the mode misses the criterion")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-read", action="store_true", help="skip roofline.measured_read_GBps (2 GiB read kernel)")
    args = ap.parse_args()

    guard = None   # (forked before torch is imported or the GPU touched: LineGuard)
    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) == 0 and args.workload == "headline"
            and args.gather_mode == "rccl" and not args.no_direct_alt):
        guard = LineGuard()

    import torch
    import torch.distributed as dist
    from auditory_amd import capi, runtime
    from auditory_amd.batch import allgather_features, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.dist_single
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        pg_timeout = datetime.timedelta(seconds=args.dist_timeout)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world, timeout=pg_timeout)
    B, K = args.batch, max(1, args.steps)
    sig_code = capi.AUD_I16 if args.sig_dtype == "i16" else capi.AUD_F32

    alone = [False]   # rank0_alone(): the collectives inside a timed region become local

    def sync_all():
        if world > 1 and not alone[0]:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def rank0_alone(fn):
        """rank 0 runs fn() while every other rank waits at the barrier behind it: the one-GPU rate of THIS run, on this node
        (the denominator of weak_scaling.x_of_1gpu)"""
        out = None
        if rank == 0:
            alone[0] = True
            try:
                out = fn()
            finally:
                alone[0] = False
        if world > 1:
            dist.barrier()
        return out

    def max_over_ranks(x):
        if world == 1 or alone[0]:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def upload_items(n, L):
        it = runtime.make_items(np.arange(n) * L, [L] * n, [0] * n)
        return torch.from_numpy(np.frombuffer(np.ascontiguousarray(it).tobytes(), np.uint8).copy()).to(dev)

    rings = {}

    def ring_for(wl, nb=None):
        nb = nb or B
        key = (wl.name, nb)
        if key not in rings:
            per_batch = nb * wl.L * (2 if args.sig_dtype == "i16" else 4)
            R = max(2, int(math.ceil(args.ring_mb * 1e6 / per_batch)))
            rings[key] = Ring(torch, wl, nb, R, rank, dev, args.sig_dtype, stereo=args.stereo,
                              distinct=64 if wl.dur_s > 1.0 else (1024 if nb > 1024 else None))
        return rings[key]

    cur = lambda: torch.cuda.current_stream(dev).cuda_stream  # noqa: E731

    def bounded_first_step(launch, what):
        """ONE eager step with the collective in it, waited for with a deadline (an event polled from the host): a peer that
        died or never reached this point leaves the collective's kernel spinning for ever -- inside a captured graph that is a
        hang nobody can attribute.  On the deadline every surviving rank prints the reason and leaves with a non-zero status;
        os._exit, because a normal exit would wait for the stuck queue."""
        launch(0, cur())
        ev = torch.cuda.Event()
        ev.record()
        t0 = time.perf_counter()
        while not ev.query():
            if time.perf_counter() - t0 > args.dist_timeout:
                print("FATAL: rank %d: the first step with %s did not complete within %.0f s (a peer missing or stuck?); "
                      "library says: %r" % (rank, what, args.dist_timeout,
                                            capi.load().aud_last_error(runtime.get_ctx(local_rank).handle)), file=sys.stderr, flush=True)
                os._exit(5)
            time.sleep(0.002)

    def timed_region(launch, n_streams, min_seconds, audio_s_per_step, extra_streams=(), reset=None, even=False):
        """K x repeats steps of launch(i, stream handle); the steps dealt over n_streams streams inside one hipGraph.
        extra_streams: streams launch() itself puts work on (forked from / joined to the capturing stream with the lanes);
        reset(): forget cross-step state (events) recorded outside the graph that is about to be captured"""
        n_streams = max(1, n_streams)
        side = [torch.cuda.Stream(dev) for _ in range(n_streams - 1)]
        GK = K * max(1, -(-args.graph_steps // K)) if args.launch == "graph" else K   # steps per replay: a multiple of K
        if even and GK % 2:   # (the direct gather's two receive slabs must keep alternating across replays)
            GK *= 2
        for i in range(args.warmup):
            launch(i, cur())
        sync_all()
        graph, launch_mode = None, "eager"
        if args.launch == "graph":
            try:
                if reset:
                    reset()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    main_s = torch.cuda.current_stream(dev)
                    for sst in side + list(extra_streams):
                        sst.wait_stream(main_s)
                    lanes = [main_s] + side
                    for i in range(GK):
                        with torch.cuda.stream(lanes[i % n_streams]):
                            launch(i, lanes[i % n_streams].cuda_stream)
                    for sst in side + list(extra_streams):
                        main_s.wait_stream(sst)
                graph.replay()
                torch.cuda.synchronize(dev)
                launch_mode = "hipGraph of %d steps" % GK + (" (%d x the %d requested)" % (GK // K, K) if GK != K else "")
            except Exception as ex:
                # capture unsupported here (CPU dry run), or refused for this region (a collective that will not be captured
                # at N > 1): the region is finished with EAGER launches in this same process -- never a re-exec of a process
                # that has touched the GPU -- and `launch` says so with the reason
                why = str(ex).splitlines()[0][:160] if str(ex) else type(ex).__name__
                print("WARNING: hipGraph capture failed (%s); timing eager launches" % why, file=sys.stderr)
                graph = None
                GK = K
                launch_mode = "eager (hipGraph capture failed: %s)" % why
                try:
                    torch.cuda.synchronize(dev)
                except Exception:     # (a capture that died half-way can leave an error behind: clear it and go on)
                    torch.cuda.synchronize(dev)
                if reset:
                    reset()

        def replay():
            if graph is not None:
                graph.replay()
            else:
                for i in range(GK):
                    launch(i, cur())

        # calibrate the number of replays: >= min_seconds of device time, the same count on every rank
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        replay()
        e1.record()
        torch.cuda.synchronize(dev)
        one = max(1e-6, e0.elapsed_time(e1) * 1e-3)
        reps = int(max_over_ranks(float(min(20000, max(1, math.ceil(min_seconds / one))))))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        sync_all()
        t0 = time.perf_counter()
        evs[0].record()
        for r in range(reps):
            replay()
            evs[r + 1].record()
        sync_all()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        per_step_us = np.array([evs[r].elapsed_time(evs[r + 1]) for r in range(reps)]) * 1e3 / GK
        steps = GK * reps
        mean_us = float(per_step_us.mean())
        return {"launch": launch_mode, "streams": n_streams, "value": round(audio_s_per_step * steps / elapsed, 1),
                "steps": steps, "repeats": reps, "graph_steps": GK, "ms_per_step": round(1e3 * elapsed / steps, 5),
                "us_per_step_device": {"mean": round(mean_us, 3), "median": round(float(np.median(per_step_us)), 3),
                                       "p10": round(float(np.percentile(per_step_us, 10)), 3),
                                       "p90": round(float(np.percentile(per_step_us, 90)), 3)}}

    def check_ring(wl, ring, touched, gout=None):
        """parity of the TIMED outputs: the first streams of ring buffer 0, plus a few streams of two other buffers"""
        osd = OracleSide(wl)
        nb = ring.B
        n0 = min(nb, 256 if wl.dur_s <= 1.0 else 32)                 # long streams: a smaller sample (the oracle is ~150 audio-s/s)
        picks = [(0, np.arange(n0))] + [(r, np.arange(0, nb, max(1, nb // 16))[:16 if wl.dur_s <= 1.0 else 4])
                                        for r in sorted({touched // 2, touched - 1} - {0})]
        got = np.concatenate([ring.mel[r].cpu().numpy()[idx] for r, idx in picks])
        if gout is None:
            ref = np.concatenate([osd.mel(ring.host_rows64(r, idx)) for r, idx in picks])
            par = strict_parity(got, ref)
        else:
            refs = [osd.mel(ring.host_rows64(r, idx), gabor=True) for r, idx in picks]
            par = strict_parity(got, np.concatenate([m for m, _ in refs]))
            par["gabor"] = strict_parity(np.concatenate([gout[r].cpu().numpy()[idx] for r, idx in picks]),
                                         np.concatenate([g for _, g in refs]))
            par["pass"] = bool(par["pass"] and par["gabor"]["pass"])
        par["checked"] = "first %d streams of ring buffer 0 + %d streams each of buffers %s" % (
            n0, len(picks[1][1]) if len(picks) > 1 else 0, [r for r, _ in picks[1:]])
        return par

    def time_mode(wl, compute, kind="mel", check=True, n_streams=None, nb=None, min_seconds=None):
        """one timed region of the fused kernel on workload wl; kind: mel | gabor (configs[3]) | full (the SndEnv loop)"""
        nb = nb or B
        n_streams = n_streams or (args.streams if args.streams > 0 else (4 if kind in ("gabor", "full") else 2))
        ring = ring_for(wl, nb)
        gabor, full = kind == "gabor", kind == "full"
        plan = wl.plan(compute, local_rank, gabor=gabor, mfcc=13 if full else 0)
        for kv in args.option:
            k, v = kv.split("=")
            plan.set_option(k, int(v))
        lib, ph = plan.lib, plan.handle
        gout = [torch.zeros((nb,) + GABOR_POOLS + (2, 8), dtype=torch.float32, device=dev) for _ in range(ring.R)] if gabor else None
        if full:  # Power / LogPower [B, H, T] and the MFCC tensors: four rotating sets (a stream reuses a set four steps later)
            f32buf = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)  # noqa: E731
            sets = [dict(pw=f32buf(nb, wl.H, wl.T), lp=f32buf(nb, wl.H, wl.T), mfcc=f32buf(nb, 13, wl.T), d1=f32buf(nb, 13, wl.T),
                         d2=f32buf(nb, 13, wl.T), en=f32buf(nb, wl.T)) for _ in range(4)]
            ws_bytes = plan.segment_workspace_bytes(nb)
            for o in sets:
                o["ws"] = torch.empty(ws_bytes + 16, dtype=torch.uint8, device=dev)

        def launch(i, st):
            r = i % ring.R
            if gabor:
                rc = lib.aud_process_batch_dev(ph, ring.sig[r].data_ptr(), sig_code, ring.items.data_ptr(), nb,
                                               ring.mel[r].data_ptr(), GABOR_POOLS[0], GABOR_POOLS[1], gout[r].data_ptr(), st)
            elif full:
                o = sets[i % 4]
                if args.tail == "fused":  # SndEnv.ProcessSegment as one call (the tail rides in the mel kernel where it can)
                    rc = lib.aud_segment_batch_dev(ph, ring.sig[r].data_ptr(), sig_code, ring.items.data_ptr(), nb,
                                                   ring.mel[r].data_ptr(), o["pw"].data_ptr(), o["lp"].data_ptr(),
                                                   o["mfcc"].data_ptr(), o["d1"].data_ptr(), o["d2"].data_ptr(), o["en"].data_ptr(),
                                                   o["ws"].data_ptr(), ws_bytes, st)
                else:                     # its two parts on the stored tensors (A/B)
                    rc = lib.aud_melspec_batch_dev(ph, ring.sig[r].data_ptr(), sig_code, ring.items.data_ptr(), nb,
                                                   ring.mel[r].data_ptr(), o["pw"].data_ptr(), o["lp"].data_ptr(), st)
                    if rc == 0:
                        rc = lib.aud_mfcc_batch_dev(ph, ring.items.data_ptr(), nb, ring.mel[r].data_ptr(), o["lp"].data_ptr(),
                                                    o["mfcc"].data_ptr(), o["d1"].data_ptr(), o["d2"].data_ptr(),
                                                    o["en"].data_ptr(), st)
            else:
                rc = lib.aud_melspec_batch_dev(ph, ring.sig[r].data_ptr(), sig_code, ring.items.data_ptr(), nb,
                                               ring.mel[r].data_ptr(), None, None, st)
            if rc != 0:
                raise RuntimeError("hot path launch: %d %s" % (rc, lib.aud_last_error(plan.ctx.handle)))

        # configs[3]: two launches by default (tile kernel, then k_gabor / k_gabor_lds by compute type); ONE launch with --option
        # item_kernel=1 (workgroup-per-item kernel: the item's mel matrix stays in LDS between the frame loop and Convolve --
        # measured slower at 256 items per launch)
        one_launch = bool(gabor and plan.info("item_kernel") == 1 and "item_kernel=1" in args.option)
        res = timed_region(launch, n_streams, args.min_seconds if min_seconds is None else min_seconds,
                           nb * (1 if alone[0] else world) * wl.dur_s)
        alg = nb * (ring.sample_bytes * wl.dur + 4 * wl.nf * wl.T)      # each sample read once + each mel value written once
        if gabor:  # the pooled on/off pairs written; the unfused path also re-reads the mel tensor
            alg += nb * 4 * GABOR_POOLS[0] * GABOR_POOLS[1] * 2 * 8 + (0 if one_launch else nb * 4 * wl.nf * wl.T)
            res["launches_per_step"] = 1 if one_launch else 2
            lds_gabor = compute == "f32" or "gabor_kernel=0" in args.option
            res["gabor_path"] = ("fused: k_melspec_w20_item (workgroup per item, mel matrix in LDS, Convolve behind one barrier)"
                                 if one_launch else "two launches: mel kernel (w20x10 tiles), then " +
                                 ("the LDS-staged gabor kernel (k_gabor_lds: the item's mel matrix copied to LDS, float32 taps through "
                                  "the scalar path)" if lds_gabor and "gabor_kernel=1" not in args.option else
                                  "k_gabor (one thread per position, float64 taps and sums as gabor.go:268-283: the float64 plan's default)"))
        if full:   # Power + LogPower and the tail's four small tensors written; the unfused tail also re-reads mel + LogPower
            alg += nb * (2 * 4 * wl.H * wl.T + 4 * (3 * 13 + 1) * wl.T)
            if args.tail != "fused":
                alg += nb * (4 * wl.nf * wl.T + 4 * wl.H * wl.T)
        res.update({"workload": wl.name, "compute": compute, "kernel": plan.kernel_name, "batch": nb,
                    "ring": {"buffers": ring.R, "input_MB": round(ring.R * nb * wl.L * ring.sample_bytes / 1e6, 1)},
                    "algorithmic_bytes_per_launch": alg,
                    "achieved_GBps": round(alg / (res["us_per_step_device"]["mean"] * 1e-6) / 1e9, 2)})
        res["frac_of_hbm_peak"] = round(res["achieved_GBps"] / HBM_PEAK_GBPS, 5)
        if check and rank == 0:
            res["parity"] = check_ring(wl, ring, min(ring.R, res["graph_steps"]), gout)
        plan.close()
        return res

    def all_ranks_ok(ok):
        """True only if every rank says so (a setup step that can fail on one rank must not leave the others in a collective)"""
        if world == 1:
            return bool(ok)
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def cfg3_region(wl, compute, total, min_seconds, mode=None):
        """BASELINE configs[2] as stated: `total` utterances per step, this rank's contiguous shard through the kernel, then
        (several ranks) the RCCL all-gather of its [nb, nf, T] slab on a second stream while the next step's kernel runs;
        two output slabs alternate, a kernel waits for the gather that last read its slab.  One rank: nothing to gather."""
        ring = ring_for(wl)
        lo, hi = shard_range(total, rank, world)
        nb = hi - lo
        # this rank's shard of the batch: ring buffers back to back (rank-seeded streams); two input copies so that
        # consecutive steps do not read the same HBM lines
        def shard_rows(n_rows):
            """shard row -> row of the ring, for the two input copies of a rank whose shard has n_rows rows (every rank lays its
            shard out by this rule over ITS ring; rank 0 replays it for the other ranks' blocks in the parity check)"""
            reps = (n_rows + B - 1) // B
            bl2 = [[(r + o * reps) % ring.R for r in range(reps)] for o in (0, 1)]
            return bl2, [np.concatenate([np.arange(B) + b * B for b in bl])[:n_rows] for bl in bl2]

        bufs, ring_row = shard_rows(nb)
        sig3 = [torch.cat([ring.sig[b] for b in bl])[:nb * wl.L].contiguous() for bl in bufs]
        items3 = upload_items(nb, wl.L)
        mel3 = [torch.empty((nb, wl.nf, wl.T), dtype=torch.float32, device=dev) for _ in range(2)]
        plan = wl.plan(compute, local_rank)
        lib, ph = plan.lib, plan.handle
        gather = world > 1 or args.dist_single
        use_streams = gather and args.dist_backend == "nccl"
        comm = torch.cuda.Stream(dev) if use_streams else None
        even = total % world == 0
        direct = host = None
        mode = mode or args.gather_mode
        if gather and mode == "direct":
            if not even:
                raise SystemExit("--gather-mode direct needs the total batch to divide evenly over the ranks")
            from auditory_amd.batch import DirectGather
            err = None
            try:
                direct = DirectGather(plan.ctx, world, rank, nb * wl.nf * wl.T)
            except Exception as ex:  # (no IPC export on this box)
                err = ex
            if not all_ranks_ok(err is None):
                if direct is not None:
                    direct.close()
                plan.close()
                raise RuntimeError("direct gather: receive buffer / IPC export failed on a rank (%s)" % err)
            try:
                direct.exchange()
            except Exception as ex:
                err = ex
            if not all_ranks_ok(err is None):
                direct.close()
                plan.close()
                raise RuntimeError("direct gather: mapping a peer's buffer failed on a rank (%s)" % err)
            recv = direct.recv(dev).view(2, total, wl.nf, wl.T)   # two receive slabs: consecutive steps alternate between them
            full = [recv[0], recv[1]]
        elif gather and mode == "host":
            # SURVEY 8(e)'s alternative: NO collective -- every rank copies its slab over its own PCIe link into its slot of one
            # host buffer all ranks map (batch.HostGather); the batch ends on the host in rank order
            if not even:
                raise SystemExit("--gather-mode host needs the total batch to divide evenly over the ranks")
            from auditory_amd.batch import HostGather
            err = None
            try:
                host = HostGather.create(world, rank, nb * wl.nf * wl.T, device=dev)
            except Exception as ex:
                err = ex
            if not all_ranks_ok(err is None):
                plan.close()
                raise RuntimeError("host gather: mapping / registering the shared buffer failed on a rank (%s)" % err)
            full = [host.view(0).view(total, wl.nf, wl.T), host.view(1).view(total, wl.nf, wl.T)]
        else:
            full = [torch.empty((total,) + tuple(mel3[0].shape[1:]), dtype=torch.float32, device=dev) for _ in range(2)] if gather else None
        # RCCL through the library's OWN communicator (aud_comm_* / aud_allgather_dev: ncclAllGather on the stream it is
        # given), not through torch's process group: a collective that torch's ProcessGroupNCCL issues inside a stream capture
        # pulls its internal stream into the capture, and its watchdog thread then queries events of that stream from the
        # side -- "operation not permitted on an event last recorded in a capturing stream" aborts the process when the
        # timing is unlucky (seen once in this round's runs).  torch.distributed stays the control plane (ids, barriers).
        own_comm = gather and direct is None and host is None and even and use_streams
        if own_comm:
            import ctypes as C
            uid = [None]
            if rank == 0:
                buf = C.create_string_buffer(128)
                plan.ctx.check(lib.aud_comm_unique_id(buf))
                uid[0] = buf.raw
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            plan.ctx.check(lib.aud_comm_init(plan.ctx.handle, world, rank, uid[0]))
        done = [None, None]
        slab_of = [0, 1]   # direct pattern: which receive slab the step with output buffer s used at its last (captured) call

        def collect(s, st):
            if direct is not None:
                slab_of[s] = direct.allgather(mel3[s].data_ptr(), nb * wl.nf * wl.T, st)   # pushes + arrival signals ...
                direct.wait(st)   # ... and the wait for every peer's: behind it the step's slab is complete HERE (like ncclAllGather)
            elif host is not None:
                host.put(mel3[s], s)      # (asynchronous device-to-host copy on the current stream: the comm stream, or the lane's)
            elif own_comm:
                plan.ctx.check(lib.aud_allgather_dev(plan.ctx.handle, mel3[s].data_ptr(), full[s].data_ptr(), nb * wl.nf * wl.T, st))
            elif even:
                dist.all_gather_into_tensor(full[s], mel3[s])
            else:
                full[s] = allgather_features(mel3[s], world, n_total=total)

        def launch(i, st):
            s = i % 2
            if use_streams and done[s] is not None:
                torch.cuda.current_stream(dev).wait_event(done[s])
            rc = lib.aud_melspec_batch_dev(ph, sig3[s].data_ptr(), sig_code, items3.data_ptr(), nb, mel3[s].data_ptr(), None, None, st)
            if rc != 0:
                raise RuntimeError("hot path launch: %d" % rc)
            if not gather:
                return
            if use_streams:
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    collect(s, comm.cuda_stream)
                    done[s] = torch.cuda.Event()
                    done[s].record()
            else:
                collect(s, st)

        def reset():
            done[0] = done[1] = None

        def launch_kernel_only(i, st):
            rc = lib.aud_melspec_batch_dev(ph, sig3[i % 2].data_ptr(), sig_code, items3.data_ptr(), nb, mel3[i % 2].data_ptr(), None, None, st)
            if rc != 0:
                raise RuntimeError("hot path launch: %d" % rc)

        kernel_only = None
        if gather:   # this rank's shard through the kernel WITHOUT the collective (two streams, like the one-GPU regions)
            kernel_only = timed_region(launch_kernel_only, 2, min(min_seconds, 0.1), total * wl.dur_s)
        if gather and world > 1:
            bounded_first_step(launch, "the direct-pattern gather" if direct is not None else "the all-gather")
        # the gather's stream handling lives in launch(): one lane, the comm stream is the second
        res = timed_region(launch, 1 if gather else 2, min_seconds, total * wl.dur_s,
                           extra_streams=[comm] if use_streams else (), reset=reset, even=direct is not None)
        if use_streams:
            torch.cuda.current_stream(dev).wait_stream(comm)
            torch.cuda.synchronize(dev)
        res.update({"workload": wl.name, "compute": compute, "kernel": plan.kernel_name, "total_batch": total,
                    "streams_this_rank": nb, "rccl_ranks": world if gather else 0,
                    "collective": (("none: every rank copies its slab over PCIe into its slot of one host buffer all ranks map "
                                    "(batch.HostGather, SURVEY 8e's host mode; two slabs alternate)" if host is not None else
                                    "direct pattern (aud_allgather_direct_dev: one device-to-device push per peer on its own "
                                    "stream with its arrival signal behind it, then a bounded wait for every peer's signal; two receive "
                                    "slabs alternate)" if direct is not None else
                                    "ncclAllGather (RCCL, on the library's own communicator: aud_comm_* / aud_allgather_dev)" if own_comm else
                                    "all_gather_into_tensor (torch.distributed)") +
                                   " of this rank's [%d, %d, %d] float32 slab, on a second stream, overlapped with the next "
                                   "step's kernel" % (nb, wl.nf, wl.T)) if gather else "none (one rank)"})
        res["shard_sizes"] = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
        if gather:
            res["gathered_shape"] = list(full[0].shape)
            # the direct pattern's floor on a fully connected xGMI node: every rank must RECEIVE world - 1 slabs, one per link
            # and per step.  AMD quotes 153.6 GB/s per link BIDIRECTIONAL, i.e. 76.8 GB/s per direction -- the figure a push
            # can use; the bound at the bidirectional figure is printed beside it (what round 4's DESIGN used)
            slab_bytes = nb * wl.nf * wl.T * 4
            res["xgmi_bound_us"] = None if world == 1 else {
                "per_step": round(slab_bytes / 76.8e9 * 1e6, 1), "assumes": "one slab of %d bytes per link and step at 76.8 GB/s per "
                "direction (153.6 GB/s per link bidirectional)" % slab_bytes, "at_153.6_GBps_per_direction": round(slab_bytes / 153.6e9 * 1e6, 1)}
            # what the links allow for configs[2] AS STATED, in the line itself: one GPU takes ~ world x kernel_us for
            # the whole batch (the kernel is linear in the items); G ranks cannot finish a step before the slower of the shard's
            # kernel and the slab's way over a link
            k_us = kernel_only["us_per_step_device"]["mean"]
            x_us = res["xgmi_bound_us"]["per_step"] if res["xgmi_bound_us"] else None
            res["scaling_bound"] = {
                "xgmi_us": x_us, "kernel_us": round(k_us, 2), "one_gpu_us_for_total_batch": round(k_us * world, 1),
                "max_x_of_1gpu": round(world * k_us / max(k_us, x_us or 0.0), 2),
                "bound_by": "kernel" if not x_us or k_us >= x_us else "xgmi",
                "note": "kernel_us: this rank's shard of %d utterances through the kernel alone, measured in this run (two streams, "
                        "no collective); xgmi_us: its %d-byte slab over one link at 76.8 GB/s per direction; the step of "
                        "configs[2] as stated cannot be shorter than the larger of the two, so %d ranks can be at most "
                        "max_x_of_1gpu times one GPU (which takes ~ ranks x kernel_us for the whole batch) whatever the collective's "
                        "implementation; the measured ratio is `value` against a 1-GPU run's also.cfg3" % (nb, slab_bytes, world)}
        if direct is not None:
            res["arrival_timeouts"] = int(max_over_ranks(float(direct.timeouts())))
            res["flags_memory"] = "fine-grained" if direct.flags_fine() == 1 else "ordinary device memory (AUD_GATHER_COARSE_FLAGS=1)"
        if host is not None:
            torch.cuda.synchronize(dev)
            sync_all()   # every rank's copies have landed (own device drained, THEN the barrier) before rank 0 reads the host buffer below
            res["pcie_bound_us"] = {"per_step": round(nb * wl.nf * wl.T * 4 / 64e9 * 1e6, 1), "assumes": "one slab per step over this rank's own PCIe Gen5 x16 link at 64 GB/s"}
        if rank == 0:  # the (gathered) tensors of the last two steps against the oracle, strict: rows of EVERY rank's block
            from auditory_amd import synth
            osd = OracleSide(wl)
            got, ref, per_rank = [], [], max(4, 48 // world)
            for r in range(world if gather else 1):
                rlo, rhi = shard_range(total, r, world)
                idx = np.arange(0, rhi - rlo, max(1, (rhi - rlo) // per_rank))[:per_rank]
                for s in (0, 1):
                    src = (full[slab_of[s] if direct is not None else s][rlo:rhi]) if gather else mel3[s]
                    got.append(src.cpu().numpy()[idx])
                    if r == 0:
                        pcm = ring.pcm[ring_row[s][idx]]
                    else:  # another rank's input streams, regenerated from that rank's seeds (Ring: rank * R * B + row)
                        pcm = np.zeros((len(idx), wl.L), np.int16)
                        for q, row in enumerate(shard_rows(rhi - rlo)[1][s][idx]):
                            pcm[q, :wl.dur] = synth.utterance_pcm(2, r * ring.R * B + int(row), wl.dur, wl.sr)
                    r64 = pcm.astype(np.float64) / 32767.0
                    ref.append(osd.mel(r64 if ring.sample_bytes == 2 else r64.astype(np.float32).astype(np.float64)))
            res["parity"] = strict_parity(np.concatenate(got), np.concatenate(ref))
            res["parity"]["checked"] = "%d streams of every rank's block (%d ranks) in each of the two %s" % (
                per_rank, world if gather else 1, "gathered tensors" if gather else "output slabs")
        if direct is not None:
            sync_all()
            direct.close()
        if host is not None:
            sync_all()   # (rank 0 unlinks the buffer: not before every rank has stopped using it)
            host.close()
        if own_comm:
            sync_all()
            plan.ctx.check(lib.aud_comm_destroy(plan.ctx.handle))
        plan.close()
        if res.get("arrival_timeouts"):   # (the same count on every rank: max over ranks)
            raise RuntimeError("direct gather: %d arrival waits ran into their poll bound (AUD_GATHER_WAIT_MS): a peer's slab "
                               "never arrived -- the numbers of this region mean nothing" % res["arrival_timeouts"])
        return res

    # ---------------------------------------------------------------------------------------------------
    stand_alone = args.workload if args.workload in ("cfg5", "cfg1", "rate48k", "n512") else ("cfg1" if args.workload == "sndenv_cfg1" else "n400")
    head_wl = Workload(stand_alone)
    kind = {"cfg4": "gabor", "sndenv": "full", "sndenv_cfg1": "full"}.get(args.workload, "mel")
    head = time_mode(head_wl, args.compute, kind=kind)
    solo = head if head["streams"] == 1 else time_mode(head_wl, args.compute, kind=kind, check=False, n_streams=1)  # the kernel alone
    if rank == 0 and "parity" in head and not head["parity"]["pass"] and not args.report_anyway:
        print("FATAL: headline mode %s/%s fails the parity criterion: %s" % (head_wl.name, args.compute, head["parity"]),
              file=sys.stderr)
        if multi:
            dist.destroy_process_group()
        raise SystemExit(3)
    modes, also, cfg3 = {}, {}, None
    extras = args.workload == "headline" and not args.only_headline
    if extras and not multi:
        a_s = args.also_seconds if args.min_seconds > 0 else 0.0
        if head["streams"] != 1:
            modes["n400_%s_1stream" % args.compute] = solo
        other = "f32" if args.compute == "f64" else "f64"
        modes["n400_" + other] = time_mode(head_wl, other, min_seconds=a_s)
        wl512 = Workload("n512")
        also["n512_" + args.compute] = time_mode(wl512, args.compute, min_seconds=a_s)     # configs[1] as worded
        modes["n512_" + other] = time_mode(wl512, other, min_seconds=a_s)
        also["cfg3"] = cfg3_region(head_wl, args.compute, args.cfg3_total, a_s)           # configs[2] on one GPU
        also["cfg4"] = time_mode(head_wl, args.compute, kind="gabor", min_seconds=a_s)     # configs[3]
        for key in list(rings):                                                           # free the 16 kHz rings first
            if key[0] != "n400" or key[1] != B:
                del rings[key]
        torch.cuda.empty_cache()
        also["cfg5"] = time_mode(Workload("cfg5"), args.compute, nb=args.cfg5_batch, min_seconds=a_s)   # configs[4]
        for key in list(rings):
            if key[0] == "cfg5":
                del rings[key]
        torch.cuda.empty_cache()
    direct_alt = one_gpu = None
    if multi and args.workload == "headline":
        if world > 1:   # the one-GPU rate of the collective-free step, rank 0 alone on the node (weak_scaling.x_of_1gpu)
            one_gpu = rank0_alone(lambda: time_mode(head_wl, args.compute, kind=kind, check=False,
                                                    min_seconds=min(args.min_seconds, args.also_seconds)))
            one_gpu = [one_gpu]
            dist.broadcast_object_list(one_gpu, src=0)
            one_gpu = one_gpu[0]
        else:
            one_gpu = head
        cfg3 = cfg3_region(head_wl, args.compute, args.cfg3_total, args.min_seconds)
        if rank == 0 and not cfg3["parity"]["pass"] and not args.report_anyway:
            print("FATAL: the gathered tensor fails the parity criterion: %s" % cfg3["parity"], file=sys.stderr)
            dist.destroy_process_group()
            raise SystemExit(3)

    # roofline.traffic: HBM bytes per launch from the PMC passes (tools/profile_bench.sh), if they were taken for this
    # kernel and batch; null otherwise (it cannot be measured inside this process)
    traffic, rocprof_us = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if pmc.get("batch") == B and pmc.get("family") == head["kernel"] and pmc.get("compute") == args.compute:
            traffic = round(float(pmc["hbm_bytes_per_launch"]), 1)
            rocprof_us = round(float(pmc["avg_duration_ns"]) / 1e3, 3)
    except (OSError, ValueError, KeyError):
        pass
    what = {"headline": "BASELINE configs[1] batch on the metric's parameters: ",
            "cfg4": "BASELINE configs[3] (configs[1] + agabor.Convolve, default FilterSet 9x9/3 x 8, [11,32,2,8] pools): ",
            "sndenv": "the whole unmodified SndEnv.ProcessSegment loop on the metric's parameters (mel + Power + LogPower "
                      "tensors + MFCC tail with deltas and Energy: SURVEY 8 f-1, f-2): ",
            "cfg1": "BASELINE configs[0] parameters (N = 1103), one 100 ms segment per item: ",
            "rate48k": "processspeech's parameters on 48 kHz audio (N = 1200: a smooth length on the any-N kernel), one 100 ms segment per item: ",
            "sndenv_cfg1": "the whole unmodified SndEnv.ProcessSegment loop on BASELINE configs[0]'s parameters (N = 1103, one 100 ms "
                           "segment per item: mel + Power + LogPower tensors + MFCC tail; the any-N kernel carries the tail): ",
            "n512": "BASELINE configs[1] as worded (512-point FFT: WinMs 32): ",
            "cfg5": "BASELINE configs[4]: "}[args.workload]
    top = cfg3 if cfg3 is not None else head      # several GPUs: configs[2] as stated is the line's value
    # the SAME kernel as ONE launch of 4096 utterances, one stream: the steady-state rate without the two-stream overlap (what the
    # committed rocprofv3 row profiles/*_b4096_* measures with counters); side by side with `pipelined` in `roofline`
    big = None
    if args.workload == "headline" and not multi and not args.only_headline and B == 256:
        big = time_mode(head_wl, args.compute, check=False, n_streams=1, nb=4096, min_seconds=args.also_seconds if args.min_seconds > 0 else 0.0)
        for key in list(rings):
            if key[1] == 4096:
                del rings[key]
        torch.cuda.empty_cache()
    # ---- roofline of the dominant kernel (the frame -> mel kernel alone: the one-stream region)
    read_gbps, read_info = (None, "skipped") if (args.no_stream_read or rank != 0) else measured_stream_read(torch, dev)
    flops = head["batch"] * head_wl.T * frame_flops(head_wl)
    solo_s = solo["us_per_step_device"]["mean"] * 1e-6
    ach_tf, peak_tf = flops / solo_s / 1e12, VECTOR_PEAK_TF[args.compute]
    # `bound`: the roof that achieved / peak / frac are priced against (the contract's "hbm"); `limited_by`: what the counters say
    # holds the kernel (vector-ALU issue in the compute type)
    any_n = head["kernel"] in ("generic", "chirp2304")   # float64 (or float32) throughout, workgroup-level transform through LDS
    roof = {"bound": "hbm", "limited_by": "lds_round_trips_and_barriers" if any_n else "valu_" + args.compute, "achieved": solo["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(solo["achieved_GBps"] / HBM_PEAK_GBPS, 5), "traffic": traffic,
            "measured_read_GBps": round(read_gbps, 1) if read_gbps else None,
            "frac_of_measured": round(solo["achieved_GBps"] / read_gbps, 5) if read_gbps else None,
            "measured_read": read_info,
            "algorithmic_flops_per_launch": round(flops), "achieved_TFLOPs": round(ach_tf, 2), "vector_peak_TFLOPs": peak_tf,
            "frac_of_%s_vector_peak" % args.compute: round(ach_tf / peak_tf, 4),
            "pipelined": {"GBps": head["achieved_GBps"], "frac": round(head["achieved_GBps"] / HBM_PEAK_GBPS, 5),
                          "frac_of_measured": round(head["achieved_GBps"] / read_gbps, 5) if read_gbps else None,
                          "frac_of_%s_vector_peak" % args.compute: round(
                              flops / (head["us_per_step_device"]["mean"] * 1e-6) / 1e12 / peak_tf, 4),
                          "us_per_launch": head["us_per_step_device"]["mean"], "streams": head["streams"]},
            "one_launch_of_4096": None if big is None else {
                "us_per_launch": big["us_per_step_device"]["mean"], "us_per_256": round(big["us_per_step_device"]["mean"] / 16.0, 3),
                "GBps": big["achieved_GBps"], "frac": round(big["achieved_GBps"] / HBM_PEAK_GBPS, 5),
                "frac_of_measured": round(big["achieved_GBps"] / read_gbps, 5) if read_gbps else None, "streams": 1,
                "note": "live, HIP events, one stream; the counter-backed rocprofv3 row of the same launch is under profiles/ (ROOFLINE.md, b4096)"},
            "kernel": "frame->FFT->power->mel (%s, %s)" % (head["kernel"], args.compute),
            "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"],
            "avg_launch_us": solo["us_per_step_device"]["mean"], "rocprofv3_avg_launch_us": rocprof_us,
            "pipelined_GBps": head["achieved_GBps"],
            "note": ("workgroup-level transform through LDS (%s): a wave waits ~half its life (SQ_WAIT_ANY) on workgroup barriers and LDS / "
                     "table round trips, not on HBM and not on vector issue; achieved / peak / frac price the kernel ALONE (one stream, %d "
                     "items per launch, HIP events) against HBM as the contract asks; pipelined = the %d-stream rate (`value`); DESIGN.md "
                     "4.3 has the account and profiles/ROOFLINE.md the rocprofv3 row" % (head["kernel"], B, head["streams"]))
                    if any_n else
                    ("float64 VALU floor: 1.1 vector instructions per algorithmic flop, v_fma_f64 at 4.4-4.6 cycles, vector ALUs ~86 %% busy "
                     "at the pipelined rate (%s of the %s vector peak with HBM at %s of 8 TB/s): not HBM-bound.  achieved / peak / frac "
                     "price the kernel ALONE (one stream, %d utterances per launch, HIP events) against HBM as the contract asks; "
                     "pipelined = the %d-stream rate (`value`); one_launch_of_4096 = steady state without the overlap; "
                     "rocprofv3_avg_launch_us / traffic = the committed rocprofv3 passes of this command (profiles/pmc_traffic.json); "
                     "DESIGN.md 4.1 has the account"
                     % ("%.2f" % (flops / (head["us_per_step_device"]["mean"] * 1e-6) / 1e12 / peak_tf), args.compute,
                        "%.2f" % (head["achieved_GBps"] / HBM_PEAK_GBPS), B, head["streams"]))
                    if args.compute == "f64" else
                    ("float32 plan: vector-issue bound as well (DESIGN.md 4.1); achieved / peak / frac price the kernel alone (one stream, "
                     "%d utterances per launch) against HBM; pipelined = the %d-stream rate" % (B, head["streams"]))}
    line = {
        "metric": METRIC, "value": top["value"], "unit": "audio-seconds/sec",
        "n_gpus": world, "steps": top["steps"], "warmup": args.warmup, "ms_per_step": top["ms_per_step"],
        "higher_is_better": True, "scaling": "strong" if cfg3 is not None else "weak", "vs_baseline": None,
        "dtype": args.compute, "epilogue": args.compute if any_n else "f32", "data": "synthetic",
        "dtype_note": ("float64 plan on the any-N route (%s): transform, power, mel sums and logarithms all in float64, rounded "
                       "to float32 once at the store; strict criterion (1e-5 on every timed element, `parity`): max scaled error %.2g"
                       % (head["kernel"], (top.get("parity") or head.get("parity") or {"max_scaled_err": float("nan")})["max_scaled_err"])
                       if args.compute == "f64" and any_n else
                       "float64 plan: FFT and real-FFT split in float64; the spectrum / mel / log epilogue is float32 behind a per-frame "
                       "power-of-two scale and passes the strict criterion (1e-5 on every timed element, `parity`) by >= 25x: max "
                       "scaled error %.2g" % (top.get("parity") or head.get("parity") or {"max_scaled_err": float("nan")})["max_scaled_err"]
                       if args.compute == "f64" else "float32 throughout (explicit opt-in AUD_FAST_F32; never the default)"),
        "steps_note": "the %d steps asked for are captured %d x into one hipGraph, replayed %d x" % (
            K, top["graph_steps"] // K, top["repeats"]),
        "config": {"workload": (("BASELINE configs[2] as stated: %d synthetic 16 kHz mono utterances of 1 s per step in total, "
                                 "contiguous shards of %d per rank, kernel + one overlapped RCCL all-gather of the [%d, %d, %d] "
                                 "float32 mel tensor; the metric's parameters: " % (args.cfg3_total, cfg3["streams_this_rank"],
                                                                                  args.cfg3_total, head_wl.nf, head_wl.T))
                                if cfg3 is not None else what) + head_wl.describe(B) +
                               ("" if kind != "mel" else ", mel output only"),
                   "batch_per_gpu": B if cfg3 is None else cfg3["streams_this_rank"], "win_samples": head_wl.N,
                   "step_samples": head_wl.S, "segment_steps": head_wl.T,
                   "n_mel": head_wl.nf, "kernel": head["kernel"], "launch": top["launch"], "streams": top["streams"],
                   "steps_requested": K, "repeats": top["repeats"], "ring": head["ring"], "options": args.option,
                   "sig_dtype": args.sig_dtype,
                   "layout": ("interleaved stereo clips: two strided work items (sig_stride 2) per clip over one buffer" if args.stereo
                              else "one contiguous mono stream per work item"),
                   "sharding": "utterances, contiguous block per rank" + ("; one all-gather per step inside `value`" if cfg3 is not None
                                                                          else "; no collective (one rank)")},
        "us_per_step_device": top["us_per_step_device"],
        "parity": top.get("parity"),
        "roofline": roof,
    }
    if cfg3 is not None:
        line["rccl_ranks"] = cfg3["rccl_ranks"]
        line["xgmi_bound_us"] = cfg3.get("xgmi_bound_us")
        if "arrival_timeouts" in cfg3:
            line["arrival_timeouts"] = cfg3["arrival_timeouts"]
            line["flags_memory"] = cfg3.get("flags_memory")
        line["collective"] = cfg3["collective"]
        line["gathered_shape"] = cfg3.get("gathered_shape")
        line["shard_sizes"] = cfg3.get("shard_sizes")
        line["no_collective"] = {k: head[k] for k in ("value", "steps", "ms_per_step", "us_per_step_device", "launch", "streams",
                                                      "batch", "parity") if k in head}
        line["no_collective"]["note"] = "the sharded step without the collective: %d utterances per rank and step (weak scaling)" % B
        line["scaling_bound"] = cfg3.get("scaling_bound")
        # the collective-free step as a first-class key: every rank its own %d utterances per step, no exchange -- and the SAME step
        # on rank 0 alone while the other ranks wait, so that the ratio is measured on this node in this run
        line["weak_scaling"] = {"value": head["value"], "unit": "audio-seconds/sec", "batch_per_gpu": B,
                                "rank0_alone_value": None if one_gpu is None else one_gpu["value"],
                                "x_of_1gpu": None if one_gpu is None else round(head["value"] / one_gpu["value"], 3),
                                "ms_per_step": head["ms_per_step"], "parity": head.get("parity"),
                                "note": "no collective in the path: the batch shards over utterances (SURVEY 8e), each rank keeps its "
                                        "features; `value` above adds the all-gather that configs[2] asks for, which the links bound "
                                        "(scaling_bound)"}
    if modes:
        line["modes"] = modes
    if also:
        line["also"] = also
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(head_wl, ring_for(head_wl).pcm)
    # the same step with the OTHER reassembly (side key `direct_gather`, never `value`): xGMI is point to point, so whether RCCL's
    # schedule or one push per peer moves the 8.5 MB slabs faster is a measurement.  It runs LAST, behind a copy of
    # the finished line held by the guard process: a fault in this optional measurement (device-to-device pushes into buffers
    # mapped from other processes: the one part of the flow no single-GPU box can rehearse) cannot take the line with it
    if cfg3 is not None and args.gather_mode == "rccl" and args.cfg3_total % world == 0 and not args.no_direct_alt:
        if guard is not None:
            guard.stash(dict(line, direct_gather={"error": "the process ended during the direct-pattern side measurement; "
                                                           "every other key of this line was measured before it"}))
        try:
            if os.environ.get("AUD_BENCH_TEST_DIE_IN_SIDE_MEASUREMENT") == "1":   # (tests/test_bench_dryrun.py: what a GPU fault does)
                os._exit(7)
            direct_alt = cfg3_region(head_wl, args.compute, args.cfg3_total, args.also_seconds, mode="direct")
        except Exception as ex:
            direct_alt = {"error": str(ex).splitlines()[0][:300]}
        line["direct_gather"] = ({k: direct_alt[k] for k in ("value", "steps", "ms_per_step", "us_per_step_device", "launch",
                                                             "collective", "parity", "arrival_timeouts", "flags_memory", "xgmi_bound_us")
                                  if k in direct_alt}
                                 if "error" not in direct_alt else direct_alt)
        if "error" not in direct_alt:
            line["direct_gather"]["note"] = (
                "like for like with ncclAllGather: every step ends, inside the graph, with the wait for every peer's arrival "
                "flag (stored by the peer behind its push, system scope), two receive slabs alternate; parity is checked on "
                "both slabs; arrival_timeouts counts waits that ran into their poll bound (must be 0)")
    if guard is not None:
        guard.done()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
