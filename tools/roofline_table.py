#!/usr/bin/env python3
"""profiles/ROOFLINE.md from the per-kernel JSONs tools/profile_bench.sh leaves under profiles/ (<tag>_kernels.json:
rocprofv3 --kernel-trace --stats average duration, FETCH_SIZE / WRITE_SIZE passes, two SQ passes).

For every shipped kernel: ALGORITHMIC bytes per launch (every input sample read once, every output value
written once; for the unfused stages also their re-reads) / rocprofv3's average duration / 8 TB/s, the HBM bytes the PMC
passes measured (FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE), algorithmic flops against
the vector peak of the compute type (78.6 TF float64 / 157.3 TF float32), and the SQ counters per wave.

usage: python tools/roofline_table.py TAG=workload[,batch] ...   e.g.  r06a=headline r06b=cfg5,1280 r06c=cfg4
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_GBPS, PEAK_TF = 8000.0, {"f64": 78.6, "f32": 157.3}
# the measured HBM-read roof (SURVEY 8d's denominator): best launch shape of tools/ubench/stream_read.hip over 2 GiB on the
# same pool (profiles/round4_stream_read.txt; bench.py measures it live as roofline.measured_read_GBps)
MEASURED_READ_GBPS = 6047.0

# workload -> samples per stream, frames per stream, mel filters, bins, kFLOP per frame (2.5 N log2 N + 3 H + 2 sum of
# the triangles' widths + nf logarithms, evaluated on the product's own mel table: bench.py frame_flops gives the same figure)
WL = {"headline": (16000, 104, 40, 201, 10.14), "n512": (16000, 104, 40, 257, 13.43), "cfg4": (16000, 104, 40, 201, 10.14),
      "sndenv": (16000, 104, 40, 201, 10.14), "cfg5": (220500, 504, 128, 1025, 63.7), "cfg1": (4410, 14, 32, 552, 30.39),
      "rate48k": (4800, 14, 32, 601, 33.35), "sndenv_cfg1": (4410, 14, 32, 552, 30.39)}


def algorithmic(fam, wl, B):
    """(bytes, flops) one launch of kernel family `fam` moves / does in workload wl at batch B"""
    dur, T, nf, H, kflop = WL[wl]
    mel_out = 4 * nf * T
    if fam in ("w20x10", "w16x16", "w64x16", "generic", "chirp2304"):
        extra = 2 * 4 * H * T if wl.startswith("sndenv") else 0   # Power + LogPower tensors
        return B * (4 * dur + mel_out + extra), B * T * kflop * 1e3
    if fam == "w20item":  # workgroup per item; cfg4: + the fused Convolve (no re-read of mel), [11, 32, 2, 8] written
        gab = wl == "cfg4"
        return (B * (4 * dur + mel_out + (4 * 11 * 32 * 16 if gab else 0)),
                B * T * kflop * 1e3 + (B * 2 * 11 * 32 * 8 * 81 if gab else 0))
    if fam == "finish":                                            # fused tail: unrounded DCT rows + per-tile Energy sums read (float64
        tiles = (T + 5) // 6                                       # workspace), mfcc / deltas / delta-deltas / Energy written
        return B * (8 * 13 * T + 8 * tiles * T + 4 * (3 * 13 + 1) * T), B * T * (2 * tiles + 8 * 13 * 2)
    if fam == "gabor":                                             # re-read mel, write [11, 32, 2, 8]
        return B * (mel_out + 4 * 11 * 32 * 16), B * 2 * 11 * 32 * 8 * 81
    if fam == "mfcc":                                              # mel + LogPower re-read, mfcc / deltas / delta-deltas / Energy written
        return B * (mel_out + 4 * H * T + 4 * (3 * 13 + 1) * T), B * T * (2 * 13 * nf + 2 * T + 8 * 13 * 2)
    return 0, 0


def main():
    rows = []
    for arg in sys.argv[1:]:
        tag, spec = arg.split("=")
        wl, *rest = spec.split(",")
        B = int(rest[0]) if rest else 256
        path = os.path.join(ROOT, "profiles", "%s_kernels.json" % tag)
        if not os.path.exists(path):
            print("missing", path, file=sys.stderr)
            continue
        for k, v in json.load(open(path))["kernels"].items():
            if v["family"] in ("kwta",) or not v.get("avg_duration_ns"):
                continue
            nbytes, flops = algorithmic(v["family"], wl, B)
            ns = v["avg_duration_ns"]
            c = v.get("counters_per_launch", {})
            waves = c.get("SQ_WAVES")
            hbm = (v["read_bytes"] or 0) + (v["write_bytes"] or 0) if v.get("read_bytes") is not None else None
            rows.append({
                "kernel": k.replace("void ", "").replace("aud::(anonymous namespace)::", "").split("(")[0], "tag": tag, "workload": wl, "batch": B,
                "avg_us": ns / 1e3, "alg_MB": nbytes / 1e6, "GBps": nbytes / ns, "frac": nbytes / ns / PEAK_GBPS,
                "frac_meas": nbytes / ns / MEASURED_READ_GBPS, "us_per_256": ns / 1e3 * 256.0 / B,
                "hbm_MB": hbm / 1e6 if hbm else None, "TF": flops / ns / 1e3 if flops else None,
                "tf_frac": flops / ns / 1e3 / PEAK_TF[v["compute"]] if flops else None,
                "valu_per_wave": c["SQ_INSTS_VALU"] / waves if waves and "SQ_INSTS_VALU" in c else None,
                "lds_cyc_per_wave": c["SQ_LDS_IDX_ACTIVE"] / waves if waves and "SQ_LDS_IDX_ACTIVE" in c else None,
                "conflict": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None,
                "wait_any": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c else None})
    fmt = lambda x, f: "—" if x is None else f % x  # noqa: E731
    out = ["# Roofline table (rocprofv3, one MI355X)", "",
           "Built by `tools/roofline_table.py` from `profiles/<tag>_kernels.json` (written by `tools/profile_bench.sh`: "
           "`rocprofv3 --kernel-trace --stats`, then separate `--pmc` passes for FETCH_SIZE, WRITE_SIZE and two SQ sets; the "
           "program directly after `--`).  Every launch is the kernel ALONE on the chip (eager launches on one stream).",
           "`frac` = algorithmic bytes per launch / rocprofv3's average duration / 8 TB/s (SURVEY 8d: every input sample read once, "
           "every output value written once; the unfused stages' re-reads counted).  `HBM MB` = FETCH_SIZE x 2 (gfx950 "
           "correction) + WRITE_SIZE per launch.  `TF` = algorithmic flops / duration against 78.6 TF (float64) / 157.3 TF "
           "(float32) vector peak.", "",
           "`of measured` = the same GB/s over the MEASURED read roof, %.0f GB/s (a plain float32 read kernel over 2 GiB, "
           "`tools/ubench/stream_read.hip`, `profiles/round4_stream_read.txt`).  `us / 256` = the duration scaled to 256 items: the "
           "row of a launch of 4096 utterances is the counter-backed steady-state rate of the headline kernel (one launch keeps "
           "every CU busy for 16 rounds of waves, so ramp and tail are 1/16 of it) -- compare with bench.py's `ms_per_step`." % MEASURED_READ_GBPS, "",
           "| kernel | workload, batch | avg us | us / 256 | algorithmic MB | GB/s | **frac of 8 TB/s** | of measured | HBM MB (PMC) | TF (of peak) | VALU / wave | "
           "LDS cycles / wave | LDS conflict share | SQ_WAIT_ANY share | profile |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| `%s` | %s, %d | %.2f | %.2f | %.2f | %.0f | **%.3f** | %.3f | %s | %s | %s | %s | %s | %s | `%s_*` |" % (
            r["kernel"], r["workload"], r["batch"], r["avg_us"], r["us_per_256"], r["alg_MB"], r["GBps"], r["frac"], r["frac_meas"],
            fmt(r["hbm_MB"], "%.2f"),
            "—" if r["TF"] is None else "%.1f (%.2f)" % (r["TF"], r["tf_frac"]), fmt(r["valu_per_wave"], "%.0f"),
            fmt(r["lds_cyc_per_wave"], "%.0f"), fmt(r["conflict"], "%.2f"), fmt(r["wait_any"], "%.2f"), r["tag"]))
    text = "\n".join(out) + "\n"
    extra = os.path.join(ROOT, "profiles", "ROOFLINE_notes.md")
    if os.path.exists(extra):
        text += "\n" + open(extra).read()
    with open(os.path.join(ROOT, "profiles", "ROOFLINE.md"), "w") as fh:
        fh.write(text)
    print(text)


if __name__ == "__main__":
    main()
