#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats output directory into a short markdown table + a JSON twin (profiles/*.md, *.json).

    python tools/summarize_profile.py <rocprof dir> <out.md> "<command line that was profiled>"

The first table is rocprofv3's own per-kernel statistics (`*kernel_stats.csv`).  The second one is computed from the
per-dispatch trace (`*kernel_trace.csv`): launches of one kernel are split by grid size and, where one (kernel, grid) holds
launches of very different problem sizes (e.g. `gae_lanes_kernel` at T = 25 and at T = 2048 share name and grid), into
duration clusters (sorted durations, a new cluster wherever one launch takes more than 3x the previous one) -- so that the
average a bench line quotes for one problem size can be read off.  bench.py reads the JSON (`in_situ_us`)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?|\w+Kernel|copyBuffer|\w+_impl)", name)
    return m.group(1) if m else name[:60]


def clusters(vals: list, ratio: float = 3.0) -> list:
    """Sorted values split wherever the next one exceeds `ratio` x the previous one."""
    out, cur = [], []
    for v in sorted(vals):
        if cur and v > ratio * cur[-1]:
            out.append(cur)
            cur = []
        cur.append(v)
    if cur:
        out.append(cur)
    return out


def main(d: str, out: str, cmd: str) -> None:
    f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    per = defaultdict(list)  # (kernel, grid threads) -> [duration us]
    for tf in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(tf)):
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            per[(short(r["Kernel_Name"]), grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    split = []
    for (k, grid), durs in per.items():
        cl = clusters(durs)
        for i, c in enumerate(cl):
            split.append(dict(kernel=k, grid=grid, cluster=i, n_clusters=len(cl), calls=len(c), avg_us=sum(c) / len(c),
                              min_us=c[0], max_us=c[-1], total_ms=sum(c) / 1e3))
    split.sort(key=lambda r: -r["total_ms"])
    with open(out, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
        o.write(f"total kernel time: {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} dispatches\n\n")
        o.write("| kernel | calls | avg us | min us | max us | total ms | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows[:25]:
            o.write(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | "
                    f"{float(r['MaxNs']) / 1e3:.2f} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |\n")
        if split:
            o.write("\n## by grid size and duration cluster (from the per-dispatch trace)\n\n"
                    "| kernel | grid threads | cluster | calls | avg us | min us | max us | total ms |\n|---|---:|---:|---:|---:|---:|---:|---:|\n")
            for r in split[:40]:
                o.write(f"| {r['kernel']} | {r['grid']} | {r['cluster'] + 1}/{r['n_clusters']} | {r['calls']} | {r['avg_us']:.2f} | "
                        f"{r['min_us']:.2f} | {r['max_us']:.2f} | {r['total_ms']:.3f} |\n")
    kern = [dict(kernel=short(r["Name"]), calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3,
                 max_us=float(r["MaxNs"]) / 1e3, total_ms=float(r["TotalDurationNs"]) / 1e6) for r in rows]
    json.dump(dict(command=cmd, total_kernel_ms=tot / 1e6, kernels=kern, by_grid=split),
              open(re.sub(r"\.md$", "", out) + ".json", "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
