#!/usr/bin/env python3
"""Largest idle gaps between consecutive kernels (and the longest kernels) of a rocprofv3 --kernel-trace output directory:
    python tools/kernel_gaps.py <dir> [...]
Used by tools/host_stall_probe.sh to show that a slow step is a hole between two kernel nodes of one graph replay."""
import csv, glob, sys
for d in sys.argv[1:]:
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    out = []
    for a, b in zip(rows, rows[1:]):
        gap = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        dur = int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
        out.append((gap, dur, a["Kernel_Name"][:60], b["Kernel_Name"][:60], (int(a["End_Timestamp"]) - t0) / 1e6))
    print(d, len(rows), "kernels")
    for g in sorted(out, key=lambda x: -x[0])[:6]:
        print("  gap %.2f ms after %s (%.1f us) before %s at t=%.1f ms" % (g[0] / 1e6, g[2], g[1] / 1e3, g[3], g[4]))
    for g in sorted(out, key=lambda x: -x[1])[:4]:
        print("  long kernel %.2f ms %s at t=%.1f ms" % (g[1] / 1e6, g[2], g[4]))
