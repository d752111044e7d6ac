#!/usr/bin/env python3
"""What the two-stream hipGraph region of bench.py looks like on the device: from a rocprofv3 --kernel-trace CSV, the
dispatches of the frame->mel kernel in launch order -- each kernel's own duration, the start-to-start interval between
consecutive ones (= the per-step time `value` is built on) and how far consecutive launches overlap.
usage: python tools/trace_overlap.py <dir with *_kernel_trace.csv> [kernel substring]"""
import csv
import glob
import os
import sys

import numpy as np


def main():
    d, key = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_melspec_w20")
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if key in r.get("Kernel_Name", ""):
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    if len(rows) < 10:
        print("no dispatches of", key, "in", d)
        return
    s = np.array([r[0] for r in rows], np.int64)
    e = np.array([r[1] for r in rows], np.int64)
    dur, gap = (e - s) / 1e3, np.diff(s) / 1e3
    # the long timed region: consecutive dispatches closer than 3 x the median interval
    med = np.median(gap)
    in_run = gap < 3 * med
    ov = (e[:-1] - s[1:]) / 1e3          # > 0: the next kernel started before this one ended
    print("%d dispatches of %s; queues/streams seen: %s" % (len(rows), key, sorted(set(r[2] for r in rows))))
    print("kernel duration          us: median %.2f  mean %.2f  p10 %.2f  p90 %.2f" % (
        np.median(dur), dur.mean(), np.percentile(dur, 10), np.percentile(dur, 90)))
    print("start-to-start interval  us: median %.2f  mean %.2f  (inside back-to-back runs: %d of %d intervals)" % (
        np.median(gap[in_run]), gap[in_run].mean(), int(in_run.sum()), len(gap)))
    print("overlap with the next launch: %.1f %% of consecutive pairs overlap; median overlap %.2f us (of %.2f us duration)" % (
        100.0 * float((ov[in_run] > 0).mean()), float(np.median(ov[in_run])), float(np.median(dur))))
    k = min(len(rows) // 2, 2000)
    print("first timestamps of a steady stretch (start us, end us, relative to the first):")
    base = s[k]
    for i in range(k, k + 8):
        print("   %9.2f %9.2f   queue %s" % ((s[i] - base) / 1e3, (e[i] - base) / 1e3, rows[i][2]))


if __name__ == "__main__":
    main()
