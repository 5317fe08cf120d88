#!/usr/bin/env python3
"""Registers, spills and occupancy of every gfx950 kernel in csrc/ (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).

    python tools/resource_usage.py [--fail-on-spill] [file.hip ...]   # default: every csrc/*.hip; kernels with spilled VGPRs first
                                                                      # --fail-on-spill: exit 1 if any kernel spills or uses scratch

A run-time branch added to a kernel that sits at the 256-register limit can spill its DEFAULT form (round 4: the published
H1 / dH2 stores of critic_rows_train_kernel cost the K1 = 384 PPO form 31 spilled registers and 1.9 us per launch until the
switch became a template parameter) -- run this after touching a hot kernel."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tianshou_marl_amd import _build  # noqa: E402


def usage(src: str) -> list:
    r = subprocess.run([_build.HIPCC, *_build.FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull],
                       capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr[-2000:])
    out, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: \s*(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            cur = dict(kernel=re.sub(r"\(anonymous namespace\)::|\(.*\)$|^void ", "", name), file=os.path.basename(src), mangled=v, src=src)
            out.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = int(v)
    return out


def scratch_ops(src: str) -> dict:
    """mangled kernel name -> number of scratch_load / scratch_store instructions in its gfx950 ISA (hipcc -S)."""
    r = subprocess.run([_build.HIPCC, *_build.FLAGS, "-S", "--cuda-device-only", "-c", src, "-o", "-"], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr[-2000:])
    out, cur = {}, None
    for line in r.stdout.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = 0
        elif line.startswith("\t.amdhsa_kernel") or line.startswith("\ts_endpgm"):
            cur = None if line.startswith("\t.amdhsa_kernel") else cur
        elif cur and re.match(r"^\s+scratch_(load|store)", line):
            out[cur] += 1
    return out


def spilling(rows: list) -> list:
    """Kernels with spilled vector registers, or whose ISA touches a scratch frame (an array the compiler could not keep in
    registers counts too) -- except `loss_kernel<0>`, whose run-time action count indexes a 16-entry local array by design (the
    stand-alone loss for hosts that own the network, any A <= 16: 528 B of scratch, no spill; the instantiation for A = 5 has none).
    A frame that is RESERVED and never touched does not count: with > 100 scalar registers spilled to vector-register lanes (the
    rollout kernels keep ~25 store pointers live) the backend sometimes leaves a 20-byte frame behind although every spill went to
    a lane -- `scratch_ops` (scratch_load / scratch_store instructions in the kernel's ISA) is what is checked for those."""
    bad = []
    for r in rows:
        if r["kernel"] == "loss_kernel<0>":
            continue
        if r.get("VGPRs Spill", 0) or (r.get("ScratchSize", 0) and r.get("scratch_ops", 1)):
            bad.append(r)
    return bad


if __name__ == "__main__":
    from concurrent.futures import ThreadPoolExecutor

    args = [a for a in sys.argv[1:] if a != "--fail-on-spill"]
    files = args or sorted(glob.glob(os.path.join(ROOT, "tianshou_marl_amd", "csrc", "*.hip")))
    with ThreadPoolExecutor(max_workers=8) as ex:
        rows = [r for rs in ex.map(usage, files) for r in rs]
    rows.sort(key=lambda r: (-r.get("VGPRs Spill", 0), -r.get("VGPRs", 0)))
    for src in sorted({r["src"] for r in rows if r.get("ScratchSize", 0)}):   # a frame: is it touched?
        ops = scratch_ops(src)
        for r in rows:
            if r["src"] == src and r.get("ScratchSize", 0):
                r["scratch_ops"] = ops.get(r["mangled"], 1)
    print(f"{'kernel':70s} {'file':22s} vgpr agpr spill scratch occ")
    for r in rows:
        print(f"{r['kernel'][:70]:70s} {r['file'][:22]:22s} {r.get('VGPRs', 0):4d} {r.get('AGPRs', 0):4d} {r.get('VGPRs Spill', 0):5d} "
              f"{r.get('ScratchSize', 0):7d} {r.get('Occupancy', 0):3d}"
              + (f"   ({r['scratch_ops']} scratch instructions)" if "scratch_ops" in r else ""))
    print(f"{sum(1 for r in rows if r.get('VGPRs Spill', 0))} of {len(rows)} kernels spill vector registers")
    if "--fail-on-spill" in sys.argv[1:] and spilling(rows):
        print("spills / scratch in: " + ", ".join(r["kernel"] for r in spilling(rows)))
        raise SystemExit(1)
