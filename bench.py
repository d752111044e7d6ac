#!/usr/bin/env python3
"""bench.py -- audio-seconds/sec of the frame->FFT->power->mel hot path on MI355X.

Workload (BASELINE.json configs[1]): batch = 256 synthetic 16 kHz mono utterances of 1 s per GPU,
WinMs 32 (N = 512-point FFT, no taper), StepMs 10 (S = 160), one segment per utterance with
BorderSteps 2 (T = 104 frames), 40 mel filters 0-8000 Hz, mel output only.  A "step" is one pass
of the hot path over one such batch, inputs already resident in HBM.  Multi-GPU: one process per
GPU, each rank owns its own 256-utterance shard (weak scaling, no data-path collective in the
timed region); the RCCL all-gather that reassembles the feature tensor is measured in a second
region and reported under "allgather".

  python bench.py --gpus 1 --steps 2000 --warmup 50
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import memguard  # noqa: E402  resident-memory ceiling (tools/memguard.py)

memguard.install()

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(oc, sig64, L, target_s=12.0, max_threads=16, chunk=8, audio_s=1.0):
    """The oracle (C float64 restatement, FFT plan cached per segment) timed on this box's host
    cores, one utterance per thread at a time (ctypes releases the GIL).

    Memory is bounded by construction: every call hands the oracle `chunk` utterances that are
    VIEWS of the resident batch (no per-thread copies), the thread count is capped at the box's
    CPU share (16 per GPU), and the sample size is capped.  (An earlier version copied
    ~1 GB per thread and took a GPU box down by exhausting host RAM.)"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, max_threads))
    n_rows = len(sig64)
    chunk = min(chunk, n_rows)
    flat = sig64.reshape(-1)                       # view of the resident [B, L] batch

    def run_chunk(first, faithful=False):
        first = first % (n_rows - chunk + 1)
        view = flat[first * L:(first + chunk) * L]  # contiguous view, no copy
        rc, mel, _ = orc.process_batch(oc.sp, oc.d, oc.m, oc.bins, oc.filt, view,
                                       np.arange(chunk) * L, np.full(chunk, L), np.zeros(chunk),
                                       faithful=faithful)
        assert rc == 0
        return chunk

    t0 = time.perf_counter()
    run_chunk(0)
    per_utt = (time.perf_counter() - t0) / chunk
    t0 = time.perf_counter()
    run_chunk(0, faithful=True)
    per_utt_faithful = (time.perf_counter() - t0) / chunk
    calls_per_thread = int(min(2000, max(1, target_s / (per_utt * chunk))))

    def worker(t):
        done = 0
        for c in range(calls_per_thread):
            done += run_chunk((t * 131 + c * chunk))
        return done

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        n = sum(ex.map(worker, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": round(n * audio_s / dt, 2), "unit": "audio-seconds/sec", "cores": cores,
            "kind": "port",
            "sample": "%d synthetic %g s utterances (the bench batch, re-used), oracle/auditory_oracle.c "
                      "float64, %d threads x %d calls x %d utterances, FFT plan cached per segment"
                      % (n, audio_s, cores, calls_per_thread, chunk),
            "one_thread_cached": round(audio_s / per_utt, 2),
            "one_thread_plan_per_frame": round(audio_s / per_utt_faithful, 2)}


_real_cpu_baseline = cpu_baseline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--win-ms", type=float, default=32.0, help="32 -> N=512 (headline), 25 -> N=400")
    ap.add_argument("--workload", choices=["cfg2", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2: BASELINE configs[1] (the judged line).  Secondary lines for BASELINE.md's table: "
                         "cfg4 = cfg2 + agabor.Convolve (default FilterSet, 4-D [11,32,2,8] pools); "
                         "cfg5 = 44.1 kHz 5 s streams, N=2048, 128 mel (use --batch 128)")
    ap.add_argument("--kwta", choices=["off", "exact", "tree"], default="off",
                    help="cfg4 only: add the k-WTA settling of the gabor tensor (SndEnv.ApplyKwta) to every step; "
                         "exact = the reference's float32 summation order, tree = fixed reduction tree")
    ap.add_argument("--sig-dtype", choices=["f32", "i16"], default="f32",
                    help="resident sample format: float32 (the metric's definition) or int16 PCM normalised on the "
                         "device (sound.go:116-141; half the input bytes)")
    ap.add_argument("--compute", choices=["f32", "f64"], default="f32")
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph",
                    help="graph: the K steps are replayed from hipGraphs of up to 50 captured steps each "
                         "(a ~5 us kernel is otherwise bound by the Python/ctypes launch path)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the steps are dealt over round-robin (each with its own output buffer). 1 = every "
                         "step waits for the previous one (default, what roofline.avg_launch_us is defined on); 2 lets "
                         "consecutive, independent batches overlap the way a double-buffered pipeline would")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="aud_plan_set_option switches for A/B runs, e.g. r16_input=1 (staged) or kernel=1 (generic)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for CPU dry runs)")
    ap.add_argument("--prewarm-s", type=float, default=0.3, help="untimed seconds of steady launches before the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import workloads as W
    from auditory_amd import capi, runtime, synth
    from auditory_amd.batch import BatchProcessor, allgather_features
    from oracle import oracle as orc  # cpu_baseline leg + table cross-check only

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

    name = "cfg2_16k_n512_nf40" if args.win_ms == 32.0 else "cfg2_16k_n400_nf40"
    assert args.win_ms in (32.0, 25.0)
    if args.workload == "cfg5":
        name = "cfg5_44k_n2048_nf128"
    oc = W.OracleCfg(orc, name)
    B, sr = args.batch, oc.sr
    dur = 5 * sr if args.workload == "cfg5" else 16000        # samples of real audio per stream
    L = (oc.full_len() + 63) // 64 * 64          # zero tail so every frame is in bounds, 64-sample pitch
    sig64, pcm16 = synth.batch(2, B, dur, sr, row_len=L, first_idx=rank * B)
    cdt = capi.AUD_F32 if args.compute == "f32" else capi.AUD_F64
    gab = dict(size=(9, 9), stride=(3, 3), gain=2.0, specs=W.DEFAULT_GABOR_SPECS) if args.workload == "cfg4" else None
    plan = W.product_plan(oc, cdt, gab, device=local_rank)
    for kv in args.option:
        k, v = kv.split("=")
        plan.set_option(k, int(v))
    bp = BatchProcessor(plan, dev)
    if args.sig_dtype == "i16":
        dsig, sig_code, sample_bytes = torch.from_numpy(pcm16).to(dev).view(-1), capi.AUD_I16, 2
    else:
        dsig, sig_code, sample_bytes = torch.from_numpy(sig64.astype(np.float32)).to(dev).view(-1), capi.AUD_F32, 4
    items = bp.upload_items(runtime.make_items(np.arange(B) * L, [L] * B, [0] * B))
    mel = torch.empty((B, oc.nf, oc.T), dtype=torch.float32, device=dev)

    # one step = one launch of the fused frame->mel kernel over the resident batch, through the C ABI
    lib, plan_h = plan.lib, plan.handle
    n_streams = max(1, args.streams)
    mels = [mel] + [torch.empty_like(mel) for _ in range(n_streams - 1)]
    gouts = [torch.zeros((B, 11, 32, 2, 8), dtype=torch.float32, device=dev) if gab else None for _ in range(n_streams)]
    gout = gouts[0]
    kw = None
    if args.kwta != "off":
        if not gab:
            raise SystemExit("--kwta needs --workload cfg4 (it settles the gabor tensor)")
        import ctypes
        from auditory_amd import kwta as kwta_mod
        kw = kwta_mod.KWTA()
        kw.Defaults()
        kw_ref = ctypes.byref(kw.c)
        kouts = [torch.empty_like(g) for g in gouts]
        kw_order = 0 if args.kwta == "exact" else 1
    side = [None] + [torch.cuda.Stream(dev) for _ in range(n_streams - 1)] if n_streams > 1 else [None]
    step_no = [0]

    def launch(buf, st):
        if gab:
            rc = lib.aud_process_batch_dev(plan_h, dsig.data_ptr(), sig_code, items.data_ptr(), B,
                                           mels[buf].data_ptr(), 11, 32, gouts[buf].data_ptr(), st)
        else:
            rc = lib.aud_melspec_batch_dev(plan_h, dsig.data_ptr(), sig_code, items.data_ptr(), B,
                                           mels[buf].data_ptr(), None, None, st)
        if rc == 0 and kw is not None:  # fresh pool state per utterance (they are independent sounds)
            rc = lib.aud_kwta_batch_dev(plan.ctx.handle, kw_ref, gouts[buf].data_ptr(), kouts[buf].data_ptr(), B,
                                        11, 32, 2, 8, 1, 1, None, kw_order, None, st)
        if rc != 0:
            raise RuntimeError("hot path launch: %d %s" % (rc, lib.aud_last_error(plan.ctx.handle)))

    def step():
        """one batch; with --streams > 1 consecutive steps go to different streams / output buffers"""
        buf = step_no[0] % n_streams
        step_no[0] += 1
        if buf == 0:
            launch(0, torch.cuda.current_stream(dev).cuda_stream)
        else:
            launch(buf, side[buf].cuda_stream)

    def fork():
        for sst in side[1:]:
            sst.wait_stream(torch.cuda.current_stream(dev))

    def join():
        for sst in side[1:]:
            torch.cuda.current_stream(dev).wait_stream(sst)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync_all()

    # K steps as n_rep replays of a hipGraph holding `per_graph` captured steps (K = n_rep * per_graph)
    launch_mode, graph, per_graph = "eager", None, 1
    if args.launch == "graph":
        per_graph = max(d for d in range(1, 51) if args.steps % d == 0)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fork()
                for _ in range(per_graph):
                    step()
                join()
            graph.replay()
            torch.cuda.synchronize(dev)
            launch_mode = "hipGraph x%d" % per_graph
        except Exception as ex:  # capture unsupported here: say so and time eager launches instead
            print("WARNING: hipGraph capture failed (%s); timing eager launches" % ex, file=sys.stderr)
            graph, per_graph = None, 1
            torch.cuda.synchronize(dev)
    n_rep = args.steps // per_graph
    # untimed: keep the device busy for ~0.3 s so that the timed region starts at steady clocks
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_s:
        if graph is not None:
            graph.replay()
        else:
            for _ in range(20):
                step()
        torch.cuda.synchronize(dev)
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                                  # same stream the kernels run on (side streams fork from / join it)
    if graph is not None:
        for _ in range(n_rep):
            graph.replay()
    else:
        fork()
        for _ in range(args.steps):
            step()
        join()
    ev1.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the timed output is the real thing (spot-check utterance 0 against the oracle)
    o = orc.process_segment(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig64[0])
    ok, msg = W.feature_close(mel[0].cpu().numpy(), o["mel_seg"], cdt, lin_axis=0)
    if not ok:
        print("WARNING: spot check vs oracle: " + msg, file=sys.stderr)

    # second region: the same step followed by the RCCL all-gather of the mel slabs
    ag = None
    if world > 1 and not args.no_allgather:
        def step0():
            launch(0, torch.cuda.current_stream(dev).cuda_stream)

        for _ in range(3):
            step0()
            allgather_features(mel, world)
        sync_all()
        k2 = max(10, args.steps // 4)
        t0 = time.perf_counter()
        for _ in range(k2):
            step0()
            full = allgather_features(mel, world)
        sync_all()
        e2 = time.perf_counter() - t0
        t = torch.tensor([e2], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e2 = float(t.item())
        ag = {"value": round(B * world * k2 / e2, 1), "unit": "audio-seconds/sec", "steps": k2,
              "ms_per_step": round(1e3 * e2 / k2, 4), "gathered_shape": list(full.shape),
              "note": "step + one ncclAllGather (RCCL) of the [B, 40, 104] f32 mel slab per rank"}

    # roofline.traffic: HBM bytes per launch from the PMC passes (tools/profile_bench.sh), if they were
    # taken for this kernel family and batch; null otherwise (it cannot be measured inside this process)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if args.workload == "cfg2" and B == 256 and ("r16" in pmc.get("kernel", "")) == (plan.kernel_name == "r16x16"):
            traffic = round(float(pmc["hbm_bytes_per_launch"]), 1)
    except (OSError, ValueError, KeyError):
        pass
    audio_s_per_step = B * world * (dur / float(sr))
    alg_bytes = B * (sample_bytes * dur + 4 * oc.nf * oc.T)  # each sample read once + each mel value written once
    if gab:  # unfused gabor: re-read the mel tensor, write the pooled on/off pairs
        alg_bytes += B * (4 * oc.nf * oc.T + 4 * 11 * 32 * 2 * 8)
    if kw is not None:  # read the gabor tensor, write the settled one
        alg_bytes += B * 2 * 4 * 11 * 32 * 2 * 8
    kern_ms = dev_ms / args.steps
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    line = {
        "metric": "audio-seconds/sec (16 kHz, 25 ms/10 ms, 40 mel) at 1/2/4/8 MI355X",
        "value": round(audio_s_per_step * args.steps / elapsed, 1),
        "unit": "audio-seconds/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.compute, "data": "synthetic",
        "config": {"workload": {"cfg2": "BASELINE configs[1]: batch=%d synthetic 16 kHz 1 s mono utterances per GPU, "
                                        "%d-pt FFT (WinMs %g), step 160, T=104 frames, 40 mel, mel only"
                                        % (B, oc.N, args.win_ms),
                                "cfg4": "BASELINE configs[3]: configs[1] (batch=%d, %d-pt FFT) + agabor.Convolve, "
                                        "default FilterSet 9x9/3 x 8 filters, [11,32,2,8] pools" % (B, oc.N),
                                "cfg5": "BASELINE configs[4]: %d mono streams of 5 s @44.1 kHz, 2048-pt FFT, step 441, "
                                        "T=504 frames, 128 mel" % B}[args.workload],
                   "batch_per_gpu": B, "win_samples": oc.N, "step_samples": oc.S,
                   "segment_steps": oc.T, "n_mel": oc.nf, "kernel": plan.kernel_name, "launch": launch_mode, "streams": n_streams,
                   "options": args.option, "kwta": args.kwta, "sig_dtype": args.sig_dtype,
                   "sharding": "utterances, contiguous block per rank"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                     "kernel": "frame->FFT->power->mel (%s)" % plan.kernel_name,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "avg_launch_us": round(kern_ms * 1e3, 3),
                     # priced against HBM as the contract asks; the static model has the float32
                     # FFT kernels vector-ALU-bound at about half of that roof
                     "expected_limiter": "valu"},
    }
    if ag:
        line["allgather"] = ag
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(oc, sig64, L, audio_s=dur / float(sr))
    if rank == 0:
        print(json.dumps(line))
    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
