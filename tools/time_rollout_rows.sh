#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace) of tools/time_rollout_rows.py's variants: 23 launches each, in order.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rm -rf gpurun_out/trr && mkdir -p gpurun_out/trr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trr -- python3 tools/time_rollout_rows.py > gpurun_out/trr/out.txt 2> gpurun_out/trr/err.txt
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/trr/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "rollout_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(("wave" if "wave" in r["Kernel_Name"] else "tile"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
names = ["tile form", "wave form", "wave form, no matrix products", "wave form, no heads / env step", "wave form, neither",
         "wave form, waves 0-3 only", "wave form, waves 0-3 only, no env step", "wave form, waves 0-3 only, no products"]
for i, n in enumerate(names):
    b = d[23 * i + 3: 23 * (i + 1)]
    if b:
        us = sum(x for _, x in b) / len(b)
        print(f"{n:34s} {b[0][0]} kernel {us:8.1f} us per launch, {us / 25:6.2f} us per vector step ({len(b)} launches)")
PY
