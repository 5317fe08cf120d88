#!/usr/bin/env python3
"""Issue-slot accounting per kernel from one rocprofv3 PMC pass: VALU instructions per MFMA, matrix-pipe busy share.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS \\
              --kernel-trace --output-format csv -d gpurun_out/pmc_issue -o i -- python3 bench.py --workload c3ppo --steps 4 --warmup 3
    python tools/pmc_issue.py gpurun_out/pmc_issue profiles/r04_pmc_issue_c3ppo.md

Why it matters (tools/probes/mfma_valu_overlap.hip): on gfx950 an f32 MFMA and the floating-point VALU instructions of another
wave of the same SIMD do NOT run side by side (times add; integer VALU overlaps about half, nothing is hidden for free) -- every
VALU instruction of a kernel is time the matrix pipe does not get.  SQ_INSTS_VALU counts MFMAs too (they issue through the VALU
port); "other VALU" below is the difference.  Counter values are summed over the device as rocprofv3 reports them:
SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (calibrated on the probe: exactly 32 cycles per v_mfma_f32_16x16x4_f32),
SQ_BUSY_CYCLES over the 32 SQ instances (8 XCDs x 4 SEs), so the matrix pipe's busy share of a launch is
MFMA_BUSY / (32 x SQ_BUSY) -- a clock-independent utilisation (the f32-MFMA peak of the roofline assumes 2.4 GHz; under
sustained MFMA load the part runs at ~1.8 GHz, profiles/r04_probe_mfma_valu_overlap.txt)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(\w+_kernel(?:<[^>(]*>)?|\w+Kernel|copyBuffer\w*|\w+_impl)", name)
    return m.group(1) if m else name[:60]


def main(d: str, out: str) -> None:
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for (k, grid), c in per.items():
        n = len(c.get("SQ_INSTS_VALU", []))
        if not n:
            continue
        avg = {name: sum(v) / len(v) for name, v in c.items()}
        mf, va = avg.get("SQ_INSTS_MFMA", 0.0), avg.get("SQ_INSTS_VALU", 0.0)
        rows.append(dict(kernel=k, grid=grid, calls=n, mfma=mf, other_valu=va - mf, salu=avg.get("SQ_INSTS_SALU", 0.0),
                         lds=avg.get("SQ_INSTS_LDS", 0.0), busy=avg.get("SQ_BUSY_CYCLES", 0.0),
                         mfma_busy=avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)))
    rows.sort(key=lambda r: -r["mfma"] * r["calls"])
    with open(out, "w") as o:
        o.write("# issue-slot accounting (rocprofv3 --pmc, per launch averages; wave-level instruction counts)\n\n")
        o.write("| kernel | grid threads | calls | MFMA | other VALU | other VALU per MFMA | SALU | LDS | matrix pipe busy |\n")
        o.write("|---|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for r in rows[:30]:
            per_m = r["other_valu"] / r["mfma"] if r["mfma"] else float("nan")
            share = r["mfma_busy"] / (32.0 * r["busy"]) if r["busy"] else float("nan")
            r["pipe_busy"], r["other_valu_per_mfma"] = share, per_m
            o.write(f"| {r['kernel']} | {r['grid']} | {r['calls']} | {r['mfma']:.0f} | {r['other_valu']:.0f} | {per_m:.2f} | {r['salu']:.0f} | "
                    f"{r['lds']:.0f} | {share:.3f} |\n")
    json.dump(dict(kernels=[r for r in rows[:30] if r["mfma"]]), open(re.sub(r"\.md$", "", out) + ".json", "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
