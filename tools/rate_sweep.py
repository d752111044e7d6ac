#!/usr/bin/env python3
"""What a caller gets at THEIR sample rate (run on the GPU box): processspeech's parameters -- WinMs 25, StepMs 10, SegmentMs 100,
BorderSteps 2, 32 mel filters 0 .. 8 kHz (examples/processspeech/processspeech.go:190-283) -- at the common audio rates, float64
plan, 256 segments of 100 ms per launch.  Per rate: the window length N = MSecToSamples(25, rate) (sound/sound.go:52-55: rounded),
the kernel the plan selected, microseconds per launch (one stream / two streams, hipGraph of --launches launches, HIP events),
nanoseconds per frame, and the strict parity check of segment 0..3 of two streams against the oracle (the checker, never timed).

  python tools/rate_sweep.py [--rates 8000,11025,...] [--win-ms 25]
"""
import argparse
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rates", default="8000,11025,16000,22050,24000,32000,44100,48000,88200,96000")
    ap.add_argument("--win-ms", type=float, default=25.0)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--compute", choices=["f32", "f64"], default="f64")
    ap.add_argument("--json", default="", help="also write the rows to this file")
    ap.add_argument("--lengths", default="", help="window lengths N instead of rates: '/'-separated, at 16 kHz (WinMs = N / 16)")
    ap.add_argument("--option", action="append", default=[], help="plan option name=value (e.g. chirp_kernel=0), repeatable")
    args = ap.parse_args()

    import torch
    import workloads as W
    from auditory_amd import capi, runtime, synth
    from oracle import oracle as orc   # parameter blocks / tables + the parity check only

    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    cdt = capi.AUD_F32 if args.compute == "f32" else capi.AUD_F64
    B = args.batch
    rows = []
    todo = ([(16000, int(n) / 16.0) for n in args.lengths.replace("/", ",").split(",")] if args.lengths else
            [(int(r), args.win_ms) for r in args.rates.replace("/", ",").split(",")])
    for sr, win_ms in todo:
        name = "sweep_%d_%g" % (sr, win_ms)
        W.CONFIGS[name] = (sr, win_ms, 10.0, 100.0, 100.0, 2, 32, 0.0, min(8000.0, sr / 2.0))
        try:
            oc = W.OracleCfg(orc, name)
        except AssertionError:
            rows.append({"rate": sr, "error": "mel table refused (a triangle wider than the [nf, nf + 2] table, SURVEY Q4)"})
            continue
        L = (oc.full_len() + 63) // 64 * 64
        try:
            plan = W.product_plan(oc, cdt)
        except capi.AuditoryError as ex:
            rows.append({"rate": sr, "N": oc.N, "error": str(ex)})
            continue
        for o in args.option:
            plan.set_option(o.split("=")[0], int(o.split("=")[1]))
        # ---- parity on two streams x four segments (host entry; the oracle is the checker)
        Lp = oc.sp.stride_samples * 4 + oc.N
        sig64, _ = synth.batch(11 + sr % 97, 2, Lp, sr)
        segs = [(r, s) for r in range(2) for s in range(4)]
        items = runtime.make_items([r * Lp for r, s in segs], [Lp] * len(segs), [s * oc.sp.stride_samples for r, s in segs])
        mel, _, _ = plan.melspec_host(sig64.ravel(), items)
        ref = np.stack([orc.process_segment(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig64[r], segment=s)["mel_seg"] for r, s in segs])
        ok, msg = W.feature_close(mel, ref, cdt, lin_axis=1)
        scaled = float(np.max(np.abs(mel - ref) / np.maximum(1.0, np.abs(ref))))
        # ---- timing: a ring of resident batches beyond the Infinity Cache, as bench.py
        R = max(2, int(np.ceil(320e6 / (B * L * 4))))
        base, _ = synth.batch(2, 32, L, sr, row_len=L)
        sig = np.tile(base.astype(np.float32), ((B + 31) // 32, 1))[:B]
        ring = [torch.from_numpy(np.roll(sig, r, axis=0)).to(dev).view(-1) for r in range(R)]
        raw = np.frombuffer(np.ascontiguousarray(runtime.make_items(np.arange(B) * L, [L] * B, [0] * B)).tobytes(), np.uint8).copy()
        d_items = torch.from_numpy(raw).to(dev)
        out = [torch.empty((B, oc.nf, oc.T), dtype=torch.float32, device=dev) for _ in range(4)]
        side = torch.cuda.Stream(dev)

        def launch(i, st):
            plan.melspec_dev(ring[i % R].data_ptr(), capi.AUD_F32, d_items.data_ptr(), B, out[i % 4].data_ptr(), 0, 0, st)

        res = {}
        for n_streams in (1, 2):
            def region():
                main_s = torch.cuda.current_stream(dev)
                if n_streams == 2:
                    side.wait_stream(main_s)
                lanes = [main_s, side][:n_streams]
                for i in range(args.launches):
                    launch(i, lanes[i % n_streams].cuda_stream)
                if n_streams == 2:
                    main_s.wait_stream(side)
            for i in range(10):
                launch(i, torch.cuda.current_stream(dev).cuda_stream)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                region()
            g.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / args.launches)
            res[n_streams] = statistics.median(ts)
        rows.append({"rate": sr, "N": oc.N, "S": oc.S, "T": oc.T, "kernel": plan.kernel_name,
                     "bluestein_L": 2304 if plan.kernel_name == "chirp2304" else plan.info("bluestein_L"),   # (the length the kernel in use transforms on)
                     "frames_per_wg": plan.info("generic_frames_per_wg") if plan.kernel_name in ("generic", "chirp2304") else None,
                     "us_one_stream": round(res[1], 2), "us_two_streams": round(res[2], 2),
                     "ns_per_frame": round(res[2] * 1e3 / (B * oc.T), 2),
                     "audio_s_per_s": round(B * 0.1 / (res[2] * 1e-6)), "parity_ok": bool(ok), "max_scaled_err": scaled,
                     "parity_note": "" if ok else msg})
        plan.close()
        del ring, d_items, out
        torch.cuda.empty_cache()
        print(json.dumps(rows[-1]), flush=True)
    print("\n| rate (Hz) | N | kernel (Bluestein L) | us per 256 segments, 1 / 2 streams | ns per frame | audio-s/s | strict parity (max scaled error) |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        if "error" in r:
            print("| %d | %s | -- | -- | -- | -- | %s |" % (r["rate"], r.get("N", "--"), r["error"]))
            continue
        print("| %d | %d | %s%s | %.1f / %.1f | %.1f | %s | %s (%.1e) |" % (
            r["rate"], r["N"], r["kernel"], " (%d)" % r["bluestein_L"] if r["bluestein_L"] else "", r["us_one_stream"],
            r["us_two_streams"], r["ns_per_frame"], "{:,}".format(r["audio_s_per_s"]).replace(",", " "),
            "ok" if r["parity_ok"] else "FAILED", r["max_scaled_err"]))
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(rows, fh, indent=1)
    return 0 if all(r.get("parity_ok", True) for r in rows) else 1


if __name__ == "__main__":
    sys.exit(main())
