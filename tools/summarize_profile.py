#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats output directory into a short markdown table (profiles/*.md)."""
import csv
import glob
import re
import sys


def short(name: str) -> str:
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?|\w+Kernel|copyBuffer|\w+_impl)", name)
    return m.group(1) if m else name[:60]


def main(d: str, out: str, cmd: str) -> None:
    f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
        o.write(f"total kernel time: {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} dispatches\n\n")
        o.write("| kernel | calls | avg us | min us | max us | total ms | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows[:25]:
            o.write(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | "
                    f"{float(r['MaxNs']) / 1e3:.2f} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
