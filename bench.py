#!/usr/bin/env python3
"""bench.py -- audio-seconds/sec of the frame -> FFT -> power -> mel hot path on MI355X.

Headline workload = the metric's own parameter set on BASELINE.json configs[1]'s batch: 256 synthetic 16 kHz mono
utterances of 1 s per GPU and step, WinMs 25 (N = 400, no taper), StepMs 10 (S = 160), one segment per utterance with
BorderSteps 2 (T = 104 frames), 40 mel filters 0-8000 Hz, mel output only.  configs[1]'s "512-pt FFT" variant (WinMs 32)
is measured in the same run and nested under "also".  A step = one launch of the fused kernel over one resident batch;
consecutive steps walk a ring of resident batches larger than the 256 MB Infinity Cache, so the input really comes
from HBM.

Headline dtype is float64: the reference computes in float64 and BASELINE.json asks for 1e-5 relative on the float32
tensors; the float32 instantiation misses that bound on a few elements per million, so it is measured
and reported beside the headline ("modes"), never as `value`.  Every mode carries a `parity` object: the timed
outputs of several ring buffers checked against the oracle under the strict criterion |d| <= 1e-5 max(1, |ref|); a
headline mode with an element past it makes the run exit non-zero without a JSON line.

The K steps asked for are captured into one hipGraph (a ~20 us kernel is otherwise bound by the Python launch path) --
repeated inside the graph until it holds at least --graph-steps (200) steps, so that the fork / join of the streams at
the ends of a replay is not what is timed -- and that graph is replayed back to back until at least --min-seconds of
device time have passed; `steps` in the JSON line is the number of steps actually timed (graph steps x repeats),
bracketed by a barrier + synchronize on both sides, max over ranks.
Inside the graph consecutive steps alternate between two streams (--streams 2, the default): the steps are independent
batches with their own output buffers, and a launch of 256 utterances is a burst of 1.5 rounds of resident waves whose
load phase and tail leave the chip half idle -- overlapping step i+1's start with step i's tail is what a double-buffered
pipeline does (27.1 -> 16.3 us per step).  `roofline` is priced on the kernel ALONE (a one-stream region of the same run).

  python bench.py                                    # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W      # one rank per GPU, RCCL

Multi-GPU: utterances are sharded in contiguous blocks (auditory_amd.batch.shard_range), `value` is the sharded step
with no collective in it (weak scaling, 256 utterances per rank); "with_allgather" repeats the steps with the path's one
collective -- the RCCL all-gather that reassembles the [B, 40, 104] feature tensor on every rank -- overlapped on a
second stream, and "cfg3" is BASELINE configs[2] as stated: 4096 utterances in total, 4096 / G per rank.
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
METRIC = "audio-seconds/sec (16 kHz, 25 ms/10 ms, 40 mel) at 1/2/4/8 MI355X"
TOL = 1e-5              # BASELINE.json north_star: 1e-5 relative on the float32 mel tensor

# name: sample rate, WinMs, StepMs, SegmentMs (= StrideMs), BorderSteps, mel filters, LoHz, HiHz, seconds of audio per stream
WORKLOADS = {
    "n400": (16000, 25.0, 10.0, 1000.0, 2, 40, 0.0, 8000.0, 1.0),     # the metric's parameters
    "n512": (16000, 32.0, 10.0, 1000.0, 2, 40, 0.0, 8000.0, 1.0),     # BASELINE configs[1] as worded ("512-pt FFT")
    "cfg5": (44100, 46.44, 10.0, 5000.0, 2, 128, 0.0, 22050.0, 5.0),  # BASELINE configs[4]
    # BASELINE configs[0]'s parameter set (processspeech defaults on the shipped 44.1 kHz WAVs: N = 1103, prime; 100 ms
    # segments of 14 steps, 32 mel): every work item is one segment -- the generic any-N kernel, a stated non-headline row
    "cfg1": (44100, 25.0, 10.0, 100.0, 2, 32, 0.0, 8000.0, 0.1),
}
GABOR_SPECS = [dict(WaveLen=2.0, Orientation=o, SigmaWidth=0.5, SigmaLength=0.5, PhaseOffset=ph, CircleEdge=True)
               for o in (0, 45, 90, 135) for ph in (0, 1.5708)]        # processspeech.go:236-252


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
        cdt = capi.AUD_F64 if compute == "f64" else capi.AUD_F32
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

    def mel(self, rows64):
        """[n, L] float64 rows -> [n, nf, T] oracle mel"""
        n, L = rows64.shape
        rc, mel, _ = self.orc.process_batch(self.sp, self.d, self.m, self.bins, self.filt, rows64.reshape(-1),
                                            np.arange(n) * L, np.full(n, L), np.zeros(n))
        assert rc == 0
        return mel


def cpu_baseline(wl, pcm, target_s=12.0, max_threads=16, chunk=8):
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
    return {"value": round(n * wl.dur_s / dt, 2), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "sample": "%d utterance passes over the first %d utterances of the bench ring (%s), oracle/auditory_oracle.c "
                      "float64, %d threads x %d calls x %d utterances, FFT plan cached per segment"
                      % (n, n_rows, wl.name, cores, calls, chunk),
            "one_thread_cached": round(wl.dur_s / per_utt, 2),
            "one_thread_plan_per_frame": round(wl.dur_s / per_utt_faithful, 2)}


_real_cpu_baseline = cpu_baseline


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

    def __init__(self, torch, wl, B, R, rank, dev, sig_dtype, need_bytes=None, stereo=False):
        from auditory_amd import runtime, synth
        self.B, self.R, self.wl = B, R, wl
        assert not stereo or B % 2 == 0
        self.pcm = np.zeros((R * B, wl.L), np.int16)                  # host copy (parity / cpu_baseline), 2 B per sample
        for i in range(R * B):
            self.pcm[i, :wl.dur] = synth.utterance_pcm(2, rank * R * B + i, wl.dur, wl.sr)
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


def main():  # noqa: C901
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--workload", choices=["headline", "cfg4", "cfg5", "cfg1", "sndenv"], default="headline",
                    help="headline: the judged line (N = 400, with the N = 512 variant under `also`).  Secondary lines for "
                         "BASELINE.md's table: cfg4 = headline + agabor.Convolve (default FilterSet, [11,32,2,8] pools); "
                         "cfg5 = 44.1 kHz 5 s streams, N = 2048, 128 mel (use --batch 1280 for >= 1 GB resident input); "
                         "sndenv = the headline parameters with everything the unmodified SndEnv.ProcessSegment loop produces "
                         "(SURVEY 8 f-1 + f-2): mel + Power + LogPower tensors + the MFCC tail (13 coefficients, deltas, "
                         "delta-deltas, Energy)")
    ap.add_argument("--compute", choices=["f64", "f32"], default="f64", help="arithmetic of the headline mode")
    ap.add_argument("--sig-dtype", choices=["f32", "i16"], default="f32",
                    help="resident sample format: float32 (the metric's definition) or int16 PCM normalised on the device "
                         "(sound.go:116-141; half the input bytes)")
    ap.add_argument("--stereo", action="store_true",
                    help="the resident streams are the channels of interleaved stereo clips (BASELINE configs[4] as worded): "
                         "two strided work items per clip over one buffer, no de-interleaving copy")
    ap.add_argument("--ring-mb", type=float, default=320.0, help="resident input ring per GPU (> the 256 MB Infinity Cache)")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="minimum device time of a timed region")
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph")
    ap.add_argument("--graph-steps", type=int, default=200,
                    help="steps one captured hipGraph holds: the requested --steps repeated until at least this many, so "
                         "that the fork/join of the streams at the two ends of a replay is not what is timed")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the steps are dealt over round-robin inside the graph (independent resident batches, "
                         "own output buffers): 1 = every step waits for the previous one; 2 (default) = consecutive launches "
                         "overlap their load burst with the previous launch's tail, as a double-buffered pipeline does "
                         "(measured 27.1 -> 16.3 us per step; 3 and 4 add nothing).  Default for --workload cfg4: 4 -- a step is "
                         "two dependent launches there and the short gabor launch fills the chip badly (measured 25.4 us per "
                         "step on 2 streams, 19.8 on 3, 19.4 on 4).  `roofline` is always taken from a 1-stream region: one "
                         "kernel alone on the chip, the duration rocprofv3 reports")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="aud_plan_set_option switches for A/B runs, e.g. kernel=2 (workgroup-tile family) or kernel=1 (generic)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for CPU dry runs)")
    ap.add_argument("--only-headline", action="store_true", help="skip the float32 / N = 512 modes and the cfg3 region")
    ap.add_argument("--cfg3-total", type=int, default=4096, help="total utterances of the configs[2] region (CPU dry runs shrink it)")
    ap.add_argument("--report-anyway", action="store_true",
                    help="secondary rows only: print the line (with parity.pass = false) when the mode misses the criterion")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from auditory_amd import capi
    from auditory_amd.batch import allgather_features, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    if args.streams <= 0:
        args.streams = 4 if args.workload in ("cfg4", "sndenv") else 2
    B, K = args.batch, max(1, args.steps)
    sig_code = capi.AUD_I16 if args.sig_dtype == "i16" else capi.AUD_F32
    gabor = args.workload == "cfg4"
    full = args.workload == "sndenv"

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    rings = {}

    def ring_for(wl):
        if wl.name not in rings:
            per_batch = B * wl.L * (2 if args.sig_dtype == "i16" else 4)
            R = max(2, int(math.ceil(args.ring_mb * 1e6 / per_batch)))
            rings[wl.name] = Ring(torch, wl, B, R, rank, dev, args.sig_dtype, stereo=args.stereo)
        return rings[wl.name]

    def time_mode(wl, compute, check=True, n_streams=None):
        """one timed region of K x repeats steps of the fused kernel; returns the per-mode result dict"""
        n_streams = max(1, n_streams or args.streams)
        side = [torch.cuda.Stream(dev) for _ in range(n_streams - 1)]
        ring = ring_for(wl)
        plan = wl.plan(compute, local_rank, gabor=gabor, mfcc=13 if full else 0)
        for kv in args.option:
            k, v = kv.split("=")
            plan.set_option(k, int(v))
        lib, ph = plan.lib, plan.handle
        gout = [torch.zeros((B, 11, 32, 2, 8), dtype=torch.float32, device=dev) for _ in range(ring.R)] if gabor else None
        if full:  # Power / LogPower [B, H, T] and the MFCC tensors: four rotating sets (a stream reuses a set four steps later)
            f32buf = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)  # noqa: E731
            sets = [dict(pw=f32buf(B, wl.H, wl.T), lp=f32buf(B, wl.H, wl.T), mfcc=f32buf(B, 13, wl.T), d1=f32buf(B, 13, wl.T),
                         d2=f32buf(B, 13, wl.T), en=f32buf(B, wl.T)) for _ in range(4)]

        def launch(i, st):
            r = i % ring.R
            if gabor:
                rc = lib.aud_process_batch_dev(ph, ring.sig[r].data_ptr(), sig_code, ring.items.data_ptr(), B,
                                               ring.mel[r].data_ptr(), 11, 32, gout[r].data_ptr(), st)
            elif full:
                o = sets[i % 4]
                rc = lib.aud_melspec_batch_dev(ph, ring.sig[r].data_ptr(), sig_code, ring.items.data_ptr(), B,
                                               ring.mel[r].data_ptr(), o["pw"].data_ptr(), o["lp"].data_ptr(), st)
                if rc == 0:
                    rc = lib.aud_mfcc_batch_dev(ph, ring.items.data_ptr(), B, ring.mel[r].data_ptr(), o["lp"].data_ptr(),
                                                o["mfcc"].data_ptr(), o["d1"].data_ptr(), o["d2"].data_ptr(), o["en"].data_ptr(), st)
            else:
                rc = lib.aud_melspec_batch_dev(ph, ring.sig[r].data_ptr(), sig_code, ring.items.data_ptr(), B,
                                               ring.mel[r].data_ptr(), None, None, st)
            if rc != 0:
                raise RuntimeError("hot path launch: %d %s" % (rc, lib.aud_last_error(plan.ctx.handle)))

        cur = lambda: torch.cuda.current_stream(dev).cuda_stream  # noqa: E731
        GK = K * max(1, -(-args.graph_steps // K)) if args.launch == "graph" else K   # steps per replay: a multiple of K
        for i in range(args.warmup):
            launch(i, cur())
        sync_all()
        graph, launch_mode = None, "eager"
        if args.launch == "graph":
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    main = torch.cuda.current_stream(dev)
                    for sst in side:
                        sst.wait_stream(main)
                    lanes = [main] + side
                    for i in range(GK):
                        launch(i, lanes[i % n_streams].cuda_stream)
                    for sst in side:
                        main.wait_stream(sst)
                graph.replay()
                torch.cuda.synchronize(dev)
                launch_mode = "hipGraph of %d steps" % GK + (" (%d x the %d requested)" % (GK // K, K) if GK != K else "")
            except Exception as ex:  # capture unsupported here (CPU dry run): say so and time eager launches
                print("WARNING: hipGraph capture failed (%s); timing eager launches" % ex, file=sys.stderr)
                graph = None
                GK = K
                torch.cuda.synchronize(dev)

        def replay():
            if graph is not None:
                graph.replay()
            else:
                for i in range(GK):
                    launch(i, cur())

        # calibrate the number of replays: >= --min-seconds of device time, the same count on every rank
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        replay()
        e1.record()
        torch.cuda.synchronize(dev)
        one = max(1e-6, e0.elapsed_time(e1) * 1e-3)
        reps = int(max_over_ranks(float(min(20000, max(1, math.ceil(args.min_seconds / one))))))
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
        audio_s = B * world * wl.dur_s
        alg = B * (ring.sample_bytes * wl.dur + 4 * wl.nf * wl.T)       # each sample read once + each mel value written once
        if gabor:  # unfused gabor: re-read the mel tensor, write the pooled on/off pairs
            alg += B * (4 * wl.nf * wl.T + 4 * 11 * 32 * 2 * 8)
        if full:   # Power + LogPower written, mel + LogPower re-read by the MFCC tail, its four small tensors written
            alg += B * (2 * 4 * wl.H * wl.T + 4 * wl.nf * wl.T + 4 * wl.H * wl.T + 4 * (3 * 13 + 1) * wl.T)
        mean_us = float(per_step_us.mean())
        res = {"workload": wl.name, "compute": compute, "kernel": plan.kernel_name, "launch": launch_mode, "streams": n_streams,
               "value": round(audio_s * steps / elapsed, 1), "steps": steps, "repeats": reps,
               "ms_per_step": round(1e3 * elapsed / steps, 5),
               "us_per_step_device": {"mean": round(mean_us, 3), "median": round(float(np.median(per_step_us)), 3),
                                      "p10": round(float(np.percentile(per_step_us, 10)), 3),
                                      "p90": round(float(np.percentile(per_step_us, 90)), 3)},
               "ring": {"buffers": ring.R, "input_MB": round(ring.R * B * wl.L * ring.sample_bytes / 1e6, 1)},
               "algorithmic_bytes_per_launch": alg,
               "achieved_GBps": round(alg / (mean_us * 1e-6) / 1e9, 2)}
        if check and rank == 0:
            # parity of the TIMED outputs: all of ring buffer 0, plus 16 streams of two other buffers
            osd = OracleSide(wl)
            touched = min(ring.R, GK)                                  # ring buffers the timed steps wrote
            n0 = min(B, 256 if wl.dur_s <= 1.0 else 24)                 # long streams: a smaller sample (the oracle is ~150 audio-s/s)
            picks = [(0, np.arange(n0))] + [(r, np.arange(0, B, max(1, B // 16))[:16 if wl.dur_s <= 1.0 else 4])
                                            for r in sorted({touched // 2, touched - 1} - {0})]
            got = np.concatenate([ring.mel[r].cpu().numpy()[idx] for r, idx in picks])
            ref = np.concatenate([osd.mel(ring.host_rows64(r, idx)) for r, idx in picks])
            res["parity"] = strict_parity(got, ref)
            res["parity"]["checked"] = "first %d streams of ring buffer 0 + %d streams each of buffers %s" % (
                n0, len(picks[1][1]) if len(picks) > 1 else 0, [r for r, _ in picks[1:]])
        plan.close()
        return res

    # ---------------------------------------------------------------------------------------------------
    head_wl = Workload(args.workload if args.workload in ("cfg5", "cfg1") else "n400")
    head = time_mode(head_wl, args.compute)
    solo = head if args.streams == 1 else time_mode(head_wl, args.compute, check=False, n_streams=1)  # the kernel alone: roofline
    if rank == 0 and "parity" in head and not head["parity"]["pass"] and not args.report_anyway:
        print("FATAL: headline mode %s/%s fails the parity criterion: %s" % (head_wl.name, args.compute, head["parity"]),
              file=sys.stderr)
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(3)
    modes, also = {}, None
    if args.workload == "headline" and world == 1 and not args.only_headline:
        if args.streams != 1:
            modes["n400_%s_1stream" % args.compute] = solo
        other = "f32" if args.compute == "f64" else "f64"
        modes["n400_" + other] = time_mode(head_wl, other)
        wl512 = Workload("n512")
        also = {"n512_" + args.compute: time_mode(wl512, args.compute), "n512_" + other: time_mode(wl512, other)}

    # ---- multi-GPU: the same steps followed by the path's one collective, overlapped on a second stream ------------
    with_ag = cfg3 = None
    if world > 1 and not args.no_allgather:
        ring = ring_for(head_wl)
        plan = head_wl.plan(args.compute, local_rank)
        lib, ph = plan.lib, plan.handle
        use_streams = args.dist_backend == "nccl"
        comm = torch.cuda.Stream(dev) if use_streams else None

        def gather_region(nb, sigs, items, mels, steps):
            """steps x (kernel over this rank's nb streams, then all-gather of its [nb, nf, T] slab on the comm stream while
            the next step's kernel runs); two output slabs alternate, a kernel waits for the gather that last read its slab"""
            done = [None, None]
            full = None

            def one(i):
                nonlocal full
                s = i % 2
                if use_streams and done[s] is not None:
                    torch.cuda.current_stream(dev).wait_event(done[s])
                rc = lib.aud_melspec_batch_dev(ph, sigs[i % len(sigs)].data_ptr(), sig_code, items.data_ptr(), nb,
                                               mels[s].data_ptr(), None, None, torch.cuda.current_stream(dev).cuda_stream)
                if rc != 0:
                    raise RuntimeError("hot path launch: %d" % rc)
                if use_streams:
                    ev = torch.cuda.Event()
                    ev.record()
                    with torch.cuda.stream(comm):
                        comm.wait_event(ev)
                        full = allgather_features(mels[s], world, n_total=nb * world)
                        done[s] = torch.cuda.Event()
                        done[s].record()
                else:
                    full = allgather_features(mels[s], world, n_total=nb * world)

            for i in range(3 if steps > 2 else 1):
                one(i)
            sync_all()
            t0 = time.perf_counter()
            for i in range(steps):
                one(i)
            sync_all()
            el = max_over_ranks(time.perf_counter() - t0)
            return el, list(full.shape)

        k2 = max(10, min(K, 200)) if args.min_seconds > 0 else 2
        el, shape = gather_region(B, ring.sig, ring.items, ring.mel[:2], k2)
        with_ag = {"value": round(B * world * head_wl.dur_s * k2 / el, 1), "unit": "audio-seconds/sec", "steps": k2,
                   "ms_per_step": round(1e3 * el / k2, 4), "gathered_shape": shape,
                   "note": "every step = kernel + one all-gather (torch.distributed all_gather_into_tensor = ncclAllGather, "
                           "RCCL) of this rank's [B, %d, %d] f32 slab, issued on a second stream and overlapped with the "
                           "next step's kernel; eager launches" % (head_wl.nf, head_wl.T)}
        if not args.only_headline:
            # BASELINE configs[2]: 4096 utterances in total, contiguous shards (auditory_amd.batch.shard_range)
            from auditory_amd import runtime
            lo, hi = shard_range(args.cfg3_total, rank, world)
            nb = hi - lo
            reps3 = (nb + B - 1) // B
            sig3 = torch.cat([ring.sig[r % ring.R] for r in range(reps3)])[:nb * head_wl.L].contiguous()
            it3 = runtime.make_items(np.arange(nb) * head_wl.L, [head_wl.L] * nb, [0] * nb)
            items3 = torch.from_numpy(np.frombuffer(np.ascontiguousarray(it3).tobytes(), np.uint8).copy()).to(dev)
            mel3 = [torch.empty((nb, head_wl.nf, head_wl.T), dtype=torch.float32, device=dev) for _ in range(2)]
            n3 = 40 if args.cfg3_total >= 1024 else 2
            el3, shape3 = gather_region(nb, [sig3], items3, mel3, n3)
            cfg3 = {"value": round(args.cfg3_total * head_wl.dur_s * n3 / el3, 1), "unit": "audio-seconds/sec", "steps": n3,
                    "ms_per_step": round(1e3 * el3 / n3, 4), "total_batch": args.cfg3_total, "streams_this_rank": nb,
                    "gathered_shape": shape3, "note": "BASELINE configs[2] as stated, strong scaling; kernel + overlapped all-gather"}
        plan.close()
    elif world == 1 and args.workload == "headline" and not args.only_headline:
        # the 1-GPU point of configs[2]'s strong-scaling curve: all 4096 utterances on this GPU, nothing to gather
        ring = ring_for(head_wl)
        from auditory_amd import runtime
        nb = args.cfg3_total
        plan = head_wl.plan(args.compute, local_rank)
        reps3 = (nb + B - 1) // B
        sig3 = torch.cat([ring.sig[r % ring.R] for r in range(reps3)])[:nb * head_wl.L].contiguous()
        it3 = runtime.make_items(np.arange(nb) * head_wl.L, [head_wl.L] * nb, [0] * nb)
        items3 = torch.from_numpy(np.frombuffer(np.ascontiguousarray(it3).tobytes(), np.uint8).copy()).to(dev)
        mel3 = torch.empty((nb, head_wl.nf, head_wl.T), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(3 if nb >= 1024 else 1):
            plan.melspec_dev(sig3.data_ptr(), sig_code, items3.data_ptr(), nb, mel3.data_ptr(), 0, 0, st)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        n3 = 40 if nb >= 1024 else 2
        for _ in range(n3):
            plan.melspec_dev(sig3.data_ptr(), sig_code, items3.data_ptr(), nb, mel3.data_ptr(), 0, 0, st)
        torch.cuda.synchronize(dev)
        el3 = time.perf_counter() - t0
        cfg3 = {"value": round(nb * head_wl.dur_s * n3 / el3, 1), "unit": "audio-seconds/sec", "steps": n3,
                "ms_per_step": round(1e3 * el3 / n3, 4), "total_batch": nb, "streams_this_rank": nb,
                "note": "BASELINE configs[2] on one GPU (the G = 1 point of its strong-scaling curve; no collective)"}
        plan.close()

    # roofline.traffic: HBM bytes per launch from the PMC passes (tools/profile_bench.sh), if they were taken for this
    # kernel and batch; null otherwise (it cannot be measured inside this process)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if pmc.get("batch") == B and pmc.get("family") == head["kernel"] and pmc.get("compute") == args.compute:
            traffic = round(float(pmc["hbm_bytes_per_launch"]), 1)
    except (OSError, ValueError, KeyError):
        pass
    line = {
        "metric": METRIC, "value": head["value"], "unit": "audio-seconds/sec",
        "n_gpus": world, "steps": head["steps"], "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.compute, "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1] batch on the metric's parameters: " if args.workload == "headline" else
                                "BASELINE configs[3] (configs[1] + agabor.Convolve, default FilterSet 9x9/3 x 8, [11,32,2,8] pools): "
                                if gabor else "the whole unmodified SndEnv.ProcessSegment loop on the metric's parameters (mel + Power + "
                                "LogPower tensors + MFCC tail with deltas and Energy: SURVEY 8 f-1, f-2): "
                                if full else "BASELINE configs[0] parameters (N = 1103), one 100 ms segment per item: "
                                if args.workload == "cfg1" else "BASELINE configs[4]: ") + head_wl.describe(B) +
                               ("" if gabor or full else ", mel output only"),
                   "batch_per_gpu": B, "win_samples": head_wl.N, "step_samples": head_wl.S, "segment_steps": head_wl.T,
                   "n_mel": head_wl.nf, "kernel": head["kernel"], "launch": head["launch"], "streams": head["streams"],
                   "steps_requested": K,
                   "repeats": head["repeats"], "ring": head["ring"], "options": args.option, "sig_dtype": args.sig_dtype,
                   "layout": ("interleaved stereo clips: two strided work items (sig_stride 2) per clip over one buffer" if args.stereo
                              else "one contiguous mono stream per work item"),
                   "sharding": "utterances, contiguous block per rank; no collective inside `value`"},
        "us_per_step_device": head["us_per_step_device"],
        "parity": head.get("parity"),
        "roofline": {"bound": "hbm", "achieved": solo["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(solo["achieved_GBps"] / HBM_PEAK_GBPS, 5), "traffic": traffic,
                     "kernel": "frame->FFT->power->mel (%s, %s)" % (head["kernel"], args.compute),
                     "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"],
                     "avg_launch_us": solo["us_per_step_device"]["mean"],
                     "pipelined_GBps": head["achieved_GBps"],
                     "note": "achieved = algorithmic bytes (every sample read once, every mel value written once) / mean device "
                             "time per launch between HIP events in a ONE-stream region (the kernel alone on the chip, kernel-to-"
                             "kernel boundary included; rocprofv3's average duration for this kernel is the same number); "
                             "pipelined_GBps = the same bytes / time per step of the %d-stream region `value` comes from; the "
                             "kernel is vector-ALU / LDS / latency bound, not HBM bound (DESIGN.md 4)" % max(1, args.streams)},
    }
    if modes:
        line["modes"] = modes
    if also:
        line["also"] = also
    if with_ag:
        line["with_allgather"] = with_ag
    if cfg3:
        line["cfg3"] = cfg3
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(head_wl, ring_for(head_wl).pcm)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
