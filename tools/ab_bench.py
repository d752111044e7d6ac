#!/usr/bin/env python3
"""Interleaved A/B timing of kernel builds / plan options in ONE process (run on the GPU box).

A variant is a library build (--libs: comma-separated tags of auditory_amd/libauditory_hip_<tag>.so, "" = the shipped
library; experimental builds come from `python -m auditory_amd.build --tag T -D<SWITCH> on a patched tree: profiles/*_experiment.patch, round6_retired_experiment_switches.patch`) and an option set.  The
variants are timed round-robin for --rounds rounds of --launches back-to-back launches each inside one hipGraph (HIP
events on the launch stream; --streams 2 deals the launches over two streams as bench.py does), in a fresh order every
round, and the per-launch median / min over rounds is printed.  Variance between processes or devices never enters.

  python tools/ab_bench.py --win-ms 25 --compute f64 --libs ,exp1
"""
import argparse
import os
import random
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--win-ms", type=float, default=25.0)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--warm", type=int, default=20, help="untimed launches per variant before the rounds")
    ap.add_argument("--compute", choices=["f32", "f64"], default="f64")
    ap.add_argument("--libs", default="", help="comma-separated build tags; the empty tag is the shipped library")
    ap.add_argument("--generic", action="store_true", help="also time the generic kernel (plan option kernel = 1)")
    ap.add_argument("--cfg", default="", help="a tests/workloads.py configuration by name instead of --win-ms (e.g. cfg1_44k_n1103_nf32)")
    ap.add_argument("--opts", default="", help="further variants per library: ';'-separated plan option sets 'name=value,name=value'")
    args = ap.parse_args()

    import torch
    import workloads as W
    from auditory_amd import capi, runtime, synth
    from oracle import oracle as orc   # parameter blocks / tables only

    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    name = {32.0: "cfg2_16k_n512_nf40", 25.0: "cfg2_16k_n400_nf40", 46.44: "cfg5_44k_n2048_nf128"}[args.win_ms]
    name = args.cfg or name
    oc = W.OracleCfg(orc, name)
    B = args.batch
    L = (oc.full_len() + 63) // 64 * 64
    R = max(2, int(np.ceil(320e6 / (B * L * 4))))   # ring of resident batches beyond the Infinity Cache, as bench.py
    sig64, _ = synth.batch(2, min(B, 256 if oc.sr == 16000 else 32), min(L, int(oc.sr * (1.0 if oc.sr == 16000 else 5.0))), oc.sr, row_len=L)
    reps = (B + sig64.shape[0] - 1) // sig64.shape[0]
    sig = np.tile(sig64.astype(np.float32), (reps, 1))[:B]
    ring = [torch.from_numpy(np.roll(sig, r, axis=0)).to(dev).view(-1) for r in range(R)]
    cdt = capi.AUD_F32 if args.compute == "f32" else capi.AUD_F64
    shipped = capi.LIB_PATH
    plans = {}
    for tag in args.libs.replace(":", ",").split(","):
        if tag:  # (the empty tag keeps whatever binding the process has: the shipped library, or a test's emulator build)
            capi.LIB_PATH = shipped.replace(".so", "_%s.so" % tag)
            capi._LIB, runtime._CTX = None, {}
        more = [(o, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in o.split(","))) for o in args.opts.split(";") if o]
        for oname, opts in [("", {})] + ([("generic", {"kernel": 1})] if args.generic else []) + more:
            p = W.product_plan(oc, cdt)
            for k, v in opts.items():
                p.set_option(k, v)
            plans["%s%s" % (tag or "shipped", " " + oname if oname else "")] = p
    raw = np.frombuffer(np.ascontiguousarray(runtime.make_items(np.arange(B) * L, [L] * B, [0] * B)).tobytes(), np.uint8).copy()
    items = torch.from_numpy(raw).to(dev)
    mel = [torch.empty((B, oc.nf, oc.T), dtype=torch.float32, device=dev) for _ in range(4)]
    side = [torch.cuda.Stream(dev) for _ in range(args.streams - 1)]

    def launch(p, i, st):
        p.melspec_dev(ring[i % R].data_ptr(), capi.AUD_F32, items.data_ptr(), B, mel[i % 4].data_ptr(), 0, 0, st)

    times = {v: [] for v in plans}
    graphs = {}
    def region(p):
        main_s = torch.cuda.current_stream(dev)
        for s in side:
            s.wait_stream(main_s)
        lanes = [main_s] + side
        for i in range(args.launches):
            launch(p, i, lanes[i % len(lanes)].cuda_stream)
        for s in side:
            main_s.wait_stream(s)

    for v, p in plans.items():            # warm every variant, then capture its launches into one hipGraph
        for i in range(args.warm):
            launch(p, i, torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                region(p)
            g.replay()
            graphs[v] = g
        except Exception as ex:  # no graph capture here (CPU dry run): eager launches
            print("no hipGraph (%s): eager launches" % ex)
            graphs[v] = None
    torch.cuda.synchronize()
    order = list(plans.items())
    rng = random.Random(7)
    for rnd in range(args.rounds):
        # a fresh order every round: the clock the chip settles at depends on what ran just before (measured: the variant
        # that follows a slow kernel reads up to 8 % slower), so no variant keeps a fixed predecessor
        rot = order[:]
        rng.shuffle(rot)
        for v, p in rot:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if graphs[v] is not None:
                graphs[v].replay()
            else:
                region(p)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1e3 / args.launches)   # us per launch (incl. kernel boundaries)
    secs = oc.sp.stride_samples / float(oc.sr)       # audio a work item stands for (the segment stride)
    alg = B * (4 * oc.sp.stride_samples + 4 * oc.nf * oc.T)
    print("workload: %s, batch %d, %s; %d rounds x %d launches on %d stream(s); algorithmic bytes/launch %.2f MB"
          % (name, B, args.compute, args.rounds, args.launches, args.streams, alg / 1e6))
    for v, ts in times.items():
        med, mn = statistics.median(ts), min(ts)
        print("%-24s kernel=%-8s lds %6d B, %d wg/CU  median %8.2f us  min %8.2f us  -> %6.3f of 8 TB/s, %7.2f M audio-s/s"
              % (v, plans[v].kernel_name, plans[v].info("lds_bytes"), plans[v].info("wgs_per_cu"), med, mn,
                 alg / (med * 1e-6) / 8e12, B * secs / med))
    for p in plans.values():
        p.close()


if __name__ == "__main__":
    main()
