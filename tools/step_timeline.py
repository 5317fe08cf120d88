#!/usr/bin/env python3
"""Timeline of the gradient steps inside one graph replay, from a rocprofv3 --kernel-trace directory:
    python tools/step_timeline.py <dir> <first kernel of a step, substring> [n_steps]
prints, for the LAST n_steps occurrences of that kernel, every kernel up to the next occurrence with its duration and the
idle gap in front of it (end of the previous kernel -> start of this one), then the averages per kernel name."""
import csv, glob, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


d, first, n = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
assert len(idx) > n + 1, "not enough occurrences"
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
period = []
for a, b in zip(idx[-n - 1:-1], idx[-n:]):
    period.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
    for i in range(a, b):
        r, p = rows[i], rows[i - 1]
        k = short(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap[k] += (int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3
        cnt[k] += 1
a, b = idx[-2], idx[-1]
print("one step (the last complete one):")
for i in range(a, b):
    r, p = rows[i], rows[i - 1]
    print("  gap %6.2f us | %-34s %8.2f us" % ((int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3, short(r["Kernel_Name"]),
                                            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("averages over the last %d steps (period start -> start: %s us):" % (n, ", ".join("%.1f" % x for x in period)))
for k in dur:
    print("  %-34s x%d  kernel %8.2f us   gap in front %6.2f us" % (k, cnt[k] // n, dur[k] / cnt[k], gap[k] / cnt[k]))
print("  sum of kernels %.1f us + gaps %.1f us per step" % (sum(dur.values()) / n, sum(gap.values()) / n))
