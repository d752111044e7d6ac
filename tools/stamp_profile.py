#!/usr/bin/env python3
"""Where a wave of the wave-autonomous kernels spends its life (run on the GPU box).

Loads the DIAGNOSTIC build of the library (auditory_amd/libauditory_hip_stamps.so, -DAUD_STAMPS: s_memtime stamps
at the phase boundaries, fenced against the scheduler and with the LDS queue drained at each stamp -- its run time
is not the product's, its SHARES are what to read), launches one batch and prints, per phase, the median / p90
cycles over all waves, plus the launch's timeline (first wave start -> last wave end) per XCD.

  python tools/stamp_profile.py --win-ms 32 --compute f64 --batch 256
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import memguard  # noqa: E402

memguard.install()

# stamps of the wave kernels: 0 at the top, 3 operands converted + frame scale, 4 first-pass DFT + twiddle, 5 transposes,
# 6 second pass, 7 split + power -> LDS, 8 epilogue + stores
MARKS = {400: [(0, 3, "0-3 tables staged, operands landed + converted, frame scale"), (3, 4, "3-4 pass-A DFT20 + twiddle"),
               (4, 5, "4-5 LDS transposes (half rows, re / im)"), (5, 6, "5-6 pass-B DFT10 x 2"),
               (6, 7, "6-7 split + power -> LDS (float32, scaled)"), (7, 8, "7-8 mel epilogue + stores")],
         512: [(0, 3, "0-3 tables staged, operands landed + converted, frame scale"), (3, 4, "3-4 pass-1 DFT16 + twiddle"),
               (4, 5, "4-5 LDS transpose re / im"), (5, 6, "5-6 pass-2 DFT16"),
               (6, 7, "6-7 split (shuffles) + power -> LDS"), (7, 8, "7-8 mel epilogue + stores")],
         2048: [(0, 3, "0-3 operands issued, landed + converted, frame scale"), (3, 4, "3-4 pass-1 DFT16 + twiddle (global table)"),
                (4, 5, "4-5 LDS transpose 1 re/im"), (5, 6, "5-6 pass-2 DFT16 + twiddle + transpose 2"),
                (6, 7, "6-7 pass-3 DFT4 + split + power -> LDS"), (7, 8, "7-8 mel epilogue + stores")]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--win-ms", type=float, default=32.0)
    ap.add_argument("--compute", choices=["f32", "f64"], default="f64")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="N > 0: instead of one launch, a hipGraph of N launches dealt over two streams as bench.py does (two "
                         "plans, each with its own stamp buffer); prints when, on the chip-wide 100 MHz counter, the waves of "
                         "the LAST launch of each stream started and ended -- the overlap the two-stream step time comes from "
                         "(rocprofv3's kernel trace serialises the launches of a graph, so it cannot show this)")
    args = ap.parse_args()

    import torch
    from auditory_amd import build, capi, runtime, synth
    lib_path = build.LIB.replace(".so", "_stamps.so")
    if not os.path.exists(lib_path):
        build.build(stamps=True)
    lib = C.CDLL(lib_path)
    for name, (res, argt) in capi.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, argt
    capi._LIB = lib          # this process only ever uses the diagnostic build
    import workloads as W
    from oracle import oracle as orc  # parameter blocks only
    from auditory_amd.batch import BatchProcessor

    dev = torch.device("cuda", 0)
    name = {32.0: "cfg2_16k_n512_nf40", 25.0: "cfg2_16k_n400_nf40", 46.44: "cfg5_44k_n2048_nf128"}[args.win_ms]
    oc = W.OracleCfg(orc, name, 5000.0 if args.win_ms == 46.44 else None)
    B = args.batch
    L = (oc.full_len() + 63) // 64 * 64
    sig64, _ = synth.batch(2, min(B, 256), L - 64, oc.sr, row_len=L)
    reps = (B + sig64.shape[0] - 1) // sig64.shape[0]
    sig = np.tile(sig64.astype(np.float32), (reps, 1))[:B]
    dsig = torch.from_numpy(sig).to(dev).view(-1)
    cdt = capi.AUD_F32 if args.compute == "f32" else capi.AUD_F64
    plan = W.product_plan(oc, cdt)
    fpw = plan.info("frames_per_wave")
    n_waves = B * ((oc.T + fpw - 1) // fpw)
    stamps = torch.zeros((n_waves, 16), dtype=torch.int64, device=dev)
    ptr = stamps.data_ptr()
    plan.set_option("stamps_lo", C.c_int32(ptr & 0xFFFFFFFF).value)
    plan.set_option("stamps_hi", C.c_int32((ptr >> 32) & 0xFFFFFFFF).value)
    bp = BatchProcessor(plan, dev)
    items = bp.upload_items(runtime.make_items(np.arange(B) * L, [L] * B, [0] * B))
    if args.pipeline > 0:
        n = args.pipeline
        plans, bufs = [plan], [stamps]
        for _ in range(n - 1):   # one plan (= one stamp buffer) per launch of the graph
            p2 = W.product_plan(oc, cdt)
            sb = torch.zeros((n_waves, 16), dtype=torch.int64, device=dev)
            p2.set_option("stamps_lo", C.c_int32(sb.data_ptr() & 0xFFFFFFFF).value)
            p2.set_option("stamps_hi", C.c_int32((sb.data_ptr() >> 32) & 0xFFFFFFFF).value)
            plans.append(p2)
            bufs.append(sb)
        mels = [torch.empty((B, oc.nf, oc.T), dtype=torch.float32, device=dev) for _ in range(2)]
        side = torch.cuda.Stream(dev)
        for p in plans:
            p.melspec_dev(dsig.data_ptr(), capi.AUD_F32, items.data_ptr(), B, mels[0].data_ptr(), 0, 0,
                          torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            main_s = torch.cuda.current_stream(dev)
            side.wait_stream(main_s)
            for i in range(n):
                st = (main_s, side)[i % 2]
                plans[i].melspec_dev(dsig.data_ptr(), capi.AUD_F32, items.data_ptr(), B, mels[i % 2].data_ptr(), 0, 0, st.cuda_stream)
            main_s.wait_stream(side)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        print("kernel %s, %s, batch %d: hipGraph of %d launches alternating between two streams: %.2f us per launch (diagnostic "
              "build: the stamps cost time; every launch has its own stamp buffer)" % (plan.kernel_name, args.compute, B, n,
                                                                                   e0.elapsed_time(e1) * 1e3 / n))
        spans = []
        for i, sb in enumerate(bufs):
            sv = sb.cpu().numpy().astype(np.int64)
            ok = sv[:, 8] > 0
            spans.append((i, sv[ok, 9].min(), np.median(sv[ok, 9]), np.median(sv[ok, 10]), sv[ok, 10].max()))
        base = min(sp[1] for sp in spans)
        print("launch  stream   first wave start   median start   median end   last wave end   (us, chip-wide 100 MHz counter)")
        for i, a0, am, bm, b1 in spans:
            print("%5d   %s     %12.2f   %12.2f   %10.2f   %12.2f" % (i, "AB"[i % 2], (a0 - base) / 100.0, (am - base) / 100.0,
                                                                 (bm - base) / 100.0, (b1 - base) / 100.0))
        order = sorted(spans, key=lambda sp: sp[1])
        ov = [max(0, order[k][4] - order[k + 1][1]) / max(1, order[k][4] - order[k][1]) for k in range(len(order) - 1)]
        print("consecutive launches (by first wave start) share the chip for %s of the earlier one's span"
              % ", ".join("%.0f %%" % (100 * x) for x in ov))
        print("mean start-to-start interval %.2f us, mean span of one launch %.2f us" % (
            (order[-1][1] - order[0][1]) / 100.0 / max(1, len(order) - 1), np.mean([(sp[4] - sp[1]) / 100.0 for sp in spans])))
        for p in plans[1:]:
            p.close()
        plan.close()
        return
    mel = torch.empty((B, oc.nf, oc.T), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(20):
        plan.melspec_dev(dsig.data_ptr(), capi.AUD_F32, items.data_ptr(), B, mel.data_ptr(), 0, 0, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan.melspec_dev(dsig.data_ptr(), capi.AUD_F32, items.data_ptr(), B, mel.data_ptr(), 0, 0, st)
    e1.record()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.int64)
    print("kernel %s, %s, batch %d: %d waves, launch %.1f us (diagnostic build: stamps cost time)"
          % (plan.kernel_name, args.compute, B, n_waves, e0.elapsed_time(e1) * 1e3))
    t = s[:, :9]
    ok = (t[:, 8] > 0)
    t = t[ok]
    life = t[:, 8] - t[:, 0]
    print("%-60s %9s %9s %9s  %6s" % ("phase (s_memtime ticks = shader cycles)", "median", "p10", "p90", "share"))
    for lo, hi, nm in MARKS[oc.N]:
        dd = t[:, hi] - t[:, lo]
        print("%-60s %9.0f %9.0f %9.0f  %5.1f%%" % (nm, np.median(dd), np.percentile(dd, 10), np.percentile(dd, 90),
                                                     100.0 * float(dd.sum()) / float(life.sum())))
    print("%-36s %9.0f %9.0f %9.0f" % ("wave lifetime", np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
    # launch timeline from the chip-wide 100 MHz counter (s_memrealtime; s_memtime is not comparable between CUs)
    xcc = s[ok, 12] & 15
    r0, r1 = s[ok, 9], s[ok, 10]
    base = r0.min()
    span = r1.max() - base
    print("launch timeline, 10 ns ticks of the 100 MHz counter from the first wave start: last start %d, last end %d (HIP events: %.1f us)"
          % (r0.max() - base, span, e0.elapsed_time(e1) * 1e3))
    dur = (r1 - r0).astype(np.float64)
    okd = dur > 50
    ghz = (t[okd, 8] - t[okd, 0]) / dur[okd] * 0.1
    print("s_memtime ticks per 10 ns of s_memrealtime over a wave's life: median %.3f, p10 %.3f, p90 %.3f GHz (the counter's rate, if it is the shader clock)"
          % (np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90)))
    nb = 16
    edges = np.linspace(0, span, nb + 1)
    starts, _ = np.histogram(r0 - base, bins=edges)
    ends, _ = np.histogram(r1 - base, bins=edges)
    mid = 0.5 * (edges[:-1] + edges[1:])
    resident = [int(((r0 - base <= m) & (r1 - base > m)).sum()) for m in mid]
    print("  sixteenths of the span | wave starts | wave ends | waves resident at the middle of the bin")
    for i in range(nb):
        print("  %5.1f us  %6d %6d %6d" % (mid[i] / 100.0, starts[i], ends[i], resident[i]))
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print("  XCC %d: %5d waves  first start %5d  last start %5d  last end %5d" % (x, m.sum(), r0[m].min() - base, r0[m].max() - base, r1[m].max() - base))
    cu = ((s[ok, 13] >> 8) & 15) | (((s[ok, 13] >> 13) & 7) << 4) | (xcc << 8)
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    print("  waves per CU: %d CUs seen, min %d, median %d, max %d" % (len(per_cu), per_cu.min(), np.median(per_cu), per_cu.max()))
    plan.close()


if __name__ == "__main__":
    main()
