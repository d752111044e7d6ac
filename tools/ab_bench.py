#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process (run on the GPU box).

Every variant is a plan option set (aud_plan_set_option) or another workload; the variants are
timed round-robin for --rounds rounds of --launches back-to-back launches each (HIP events on the
launch stream), and the per-launch median / min over rounds is printed.  Variance between processes
or devices never enters the comparison.

  python tools/ab_bench.py                       # N=512: r16 direct / r16 staged / generic
  python tools/ab_bench.py --win-ms 25           # N=400: r25x8 / generic
  python tools/ab_bench.py --batch 4096          # a grid that fills the chip several times over
"""
import argparse
import random
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--win-ms", type=float, default=32.0)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--warm", type=int, default=20, help="untimed launches per variant before the rounds")
    ap.add_argument("--compute", choices=["f32", "f64"], default="f32")
    args = ap.parse_args()

    import torch
    import workloads as W
    from auditory_amd import capi, runtime, synth
    from auditory_amd.batch import BatchProcessor
    from oracle import oracle as orc   # parameter blocks / tables only

    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    name = {32.0: "cfg2_16k_n512_nf40", 25.0: "cfg2_16k_n400_nf40"}[args.win_ms]
    oc = W.OracleCfg(orc, name)
    B = args.batch
    L = (oc.full_len() + 63) // 64 * 64
    sig64, _ = synth.batch(2, min(B, 256), 16000, oc.sr, row_len=L)
    reps = (B + sig64.shape[0] - 1) // sig64.shape[0]
    sig = np.tile(sig64.astype(np.float32), (reps, 1))[:B]
    dsig = torch.from_numpy(sig).to(dev).view(-1)
    cdt = capi.AUD_F32 if args.compute == "f32" else capi.AUD_F64
    variants = {"w20 (default)": {}, "w20 one tile per wave": {"wave_grid": 0}, "w20 persistent": {"wave_grid": 1},
                "w20 persistent, late prefetch": {"wave_grid": 1, "wave_variant": 0},
                "w25 (25 x 8 geometry)": {"n400_geometry": 25}, "w25 one tile per wave": {"n400_geometry": 25, "wave_grid": 0},
                "w25 persistent": {"n400_geometry": 25, "wave_grid": 1},
                "w20, no xcd remap": {"xcd_remap": 0}, "r25 tile kernel": {"kernel": 2}, "generic": {"kernel": 1}}
    if args.win_ms == 32.0:
        variants = {"w16 (default)": {}, "w16 one tile per wave": {"wave_grid": 0}, "w16 persistent var 2": {"wave_grid": 1}, "w16 persistent prefetch": {"wave_grid": 1, "wave_variant": 0},
                    "w16, no xcd remap": {"xcd_remap": 0},
                    "r16 direct": {"r16_input": 0}, "r16 direct, no xcd remap": {"r16_input": 0, "xcd_remap": 0}, "r16 direct x2": {"r16_input": 0, "r16_tiles": 2},
                    "r16 staged": {"r16_input": 1}, "r16 direct mfma-mel": {"r16_input": 0, "r16_mel": 1},
                    "r16 direct x2 mfma-mel": {"r16_input": 0, "r16_tiles": 2, "r16_mel": 1}, "generic": {"kernel": 1}}
    plans = {}
    for vname, opts in variants.items():
        p = W.product_plan(oc, cdt)
        try:
            for k, v in opts.items():
                p.set_option(k, v)
        except capi.AuditoryError as e:
            print("skip %s: %s" % (vname, e))
            p.close()
            continue
        plans[vname] = p
    bp = BatchProcessor(next(iter(plans.values())), dev)
    items = bp.upload_items(runtime.make_items(np.arange(B) * L, [L] * B, [0] * B))
    mel = torch.empty((B, oc.nf, oc.T), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def launch(p):
        p.melspec_dev(dsig.data_ptr(), capi.AUD_F32, items.data_ptr(), B, mel.data_ptr(), 0, 0, stream)

    times = {v: [] for v in plans}
    graphs = {}
    for v, p in plans.items():            # warm every variant, then capture its launches into one hipGraph
        for _ in range(args.warm):
            launch(p)
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                s_cap = torch.cuda.current_stream(dev).cuda_stream
                for _ in range(args.launches):
                    p.melspec_dev(dsig.data_ptr(), capi.AUD_F32, items.data_ptr(), B, mel.data_ptr(), 0, 0, s_cap)
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
                for _ in range(args.launches):
                    launch(p)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1e3 / args.launches)   # us per launch (incl. kernel boundaries)
    alg = B * (4 * 16000 + 4 * oc.nf * oc.T)
    print("workload: %s, batch %d, %s; %d rounds x %d launches; algorithmic bytes/launch %.2f MB"
          % (name, B, args.compute, args.rounds, args.launches, alg / 1e6))
    for v, ts in times.items():
        med, mn = statistics.median(ts), min(ts)
        print("%-28s kernel=%-9s lds %6d B, %d wg/CU  median %8.2f us  min %8.2f us  -> %6.2f TB/s algorithmic, %7.1f M audio-s/s"
              % (v, plans[v].kernel_name, plans[v].info("lds_bytes"), plans[v].info("wgs_per_cu"), med, mn,
                 alg / (med * 1e-6) / 1e12, B / med))
    for p in plans.values():
        p.close()


if __name__ == "__main__":
    main()
