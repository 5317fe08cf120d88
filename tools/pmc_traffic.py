#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic

Units and corrections (MI355X_MICROARCH.md, "HBM" / "rocprofv3 PMC slots"): both counters are in KiB of memory-side
(fabric) requests of the L2s, Infinity-Cache hits included.  On gfx950 FETCH_SIZE tallies 128-B read requests at 64 B:
a wide coalesced streaming read reports exactly half its bytes, so reads are doubled ("fetch_x2").  The guide calibrates
that factor for 16-B-per-lane loads only; the kernels here load 4 B per lane, so the factor is re-calibrated on kernels
of this very access pattern whose byte count is known and which have no reuse (the GAE scan and the stand-alone loss
kernel at 819 200 samples, both run by bench.py's roofline grid): see "calibration" in the output.
Writes are taken as reported (exact for streaming stores per the guide).
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(\w+_kernel(?:<[^>(]*>)?|\w+Kernel|copyBuffer\w*|\w+_impl)", name)
    return m.group(1) if m else name[:60]


def clusters(vals: list, ratio: float = 3.0) -> list:
    """Sorted values split wherever the next one exceeds `ratio` x the previous one: launches of one (kernel, grid) at very
    different problem sizes (gae_lanes_kernel at T = 25 and T = 2048 share name and grid) are reported apart."""
    out, cur = [], []
    for v in sorted(vals):
        if cur and v > ratio * max(cur[-1], 1.0):
            out.append(cur)
            cur = []
        cur.append(v)
    if cur:
        out.append(cur)
    return out


def load(d: str, counter: str):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    per = defaultdict(list)  # (kernel, grid) -> [KiB]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return per


def split_sizes(fetch: dict, write: dict):
    """(kernel, grid) -> (kernel, grid, cluster): only where BOTH passes show the same number (> 1) of value clusters (the two
    passes are separate runs of one command, so the k-th smallest problem size of one is the k-th smallest of the other)."""
    f2, w2 = {}, {}
    for key in set(fetch) | set(write):
        cf, cw = clusters(fetch.get(key, [])), clusters(write.get(key, []))
        if len(cf) == len(cw) and len(cf) > 1:
            for i, (a, b) in enumerate(zip(cf, cw)):
                f2[key + (i,)], w2[key + (i,)] = a, b
        else:
            if key in fetch:
                f2[key + (0,)] = fetch[key]
            if key in write:
                w2[key + (0,)] = write[key]
    return f2, w2


def main(d_fetch: str, d_write: str, out: str) -> None:
    fetch, write = split_sizes(load(d_fetch, "FETCH_SIZE"), load(d_write, "WRITE_SIZE"))
    rows = []
    for key in sorted(set(fetch) | set(write)):
        if not any(t in key[0] for t in ("gae_lanes", "loss_kernel", "ppo_update", "rollout", "adam", "policy_forward",
                                        "adv_stats", "finalize", "ppo_actor_rows", "actor_rows64", "ppo_critic_rows", "reduce_slabs",
                                        "critic_rows", "critic_dw1")):
            continue
        f = fetch.get(key, [])
        w = write.get(key, [])
        rows.append(dict(kernel=key[0], grid=key[1], size_cluster=key[2], launches=max(len(f), len(w)),
                         fetch_KiB=sum(f) / len(f) if f else None, write_KiB=sum(w) / len(w) if w else None))
    # calibration of the read factor on known streaming byte counts (4 B per lane loads)
    known = {  # (kernel substring, grid threads) -> (read bytes, write bytes)
        ("gae_lanes", 32768 * 4): (14 * 819200, 8 * 819200),     # 64 lanes x W=4 waves per 64-lane column
        ("loss_kernel", 819200): (40 * 819200, 24 * 819200),
    }
    calib = []
    for r in rows:
        for (name, grid), (rd, wr) in known.items():
            if name in r["kernel"] and r["grid"] == grid and r["fetch_KiB"] and 0.25 < rd / (r["fetch_KiB"] * 1024) < 8:
                calib.append(dict(kernel=r["kernel"], grid=grid, algorithmic_read_B=rd, fetch_reported_B=r["fetch_KiB"] * 1024,
                                  read_factor=rd / (r["fetch_KiB"] * 1024), algorithmic_write_B=wr,
                                  write_reported_B=(r["write_KiB"] or 0) * 1024))
    # the guide's gfx950 correction (x2) is applied; the calibration rows are the check that it also holds for the
    # 4-B-per-lane loads of these kernels (expected: both factors within a few % of 2.0)
    factor = 2.0
    for r in rows:
        rd = (r["fetch_KiB"] or 0.0) * 1024
        r["read_B_x2"] = 2.0 * rd
        r["read_B_calibrated"] = factor * rd
        r["write_B"] = (r["write_KiB"] or 0.0) * 1024
        r["traffic_B"] = r["read_B_calibrated"] + r["write_B"]
    res = dict(unit="bytes per launch", read_factor_used=factor,
               note="FETCH_SIZE/WRITE_SIZE in KiB from separate --pmc passes; reads scaled by the calibrated factor "
                    "(guide: x2 for wide streaming reads on gfx950), writes as reported", calibration=calib, kernels=rows)
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as o:
        o.write("# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)\n\n")
        o.write(f"read factor used: {factor:.3f} (calibrated on {len(calib)} known streaming kernels; the guide's gfx950 correction is 2.0)\n\n")
        for c in calib:
            o.write(f"- calibration `{c['kernel']}` grid {c['grid']}: algorithmic read {c['algorithmic_read_B'] / 1e6:.2f} MB, "
                    f"FETCH_SIZE {c['fetch_reported_B'] / 1e6:.2f} MB -> factor {c['read_factor']:.3f}; algorithmic write "
                    f"{c['algorithmic_write_B'] / 1e6:.2f} MB, WRITE_SIZE {c['write_reported_B'] / 1e6:.2f} MB\n")
        o.write("\n| kernel | grid (threads) | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | read B (calibrated) | write B | traffic B |\n")
        o.write("|---|---:|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            o.write(f"| {r['kernel']} | {r['grid']} | {r['launches']} | {r['fetch_KiB'] or 0:.1f} | {r['write_KiB'] or 0:.1f} | "
                    f"{r['read_B_calibrated']:.0f} | {r['write_B']:.0f} | {r['traffic_B']:.0f} |\n")
    print(open(out + ".md").read())


if __name__ == "__main__":
    main(*sys.argv[1:4])
