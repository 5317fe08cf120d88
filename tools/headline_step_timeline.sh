#!/usr/bin/env bash
# (GPU box, from the repo root) rocprofv3 kernel trace of a few steps of the headline job -> one whole step (rollout -> next
# rollout) kernel by kernel with the idle gap in front of each; the 18 (update, adam) pairs are summarised.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --no-cpu-baseline --no-c3-grid --steps 12 --warmup 5 > /dev/null 2> gpurun_out/tl.err
python - <<'PY'
import csv, glob, re
rows=[]
for f in glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def short(n):
    m=re.search(r"(\w+_kernel)",n); return m.group(1) if m else n[:50]
idx=[i for i,r in enumerate(rows) if "rollout_kernel" in r["Kernel_Name"]]
for a,b in zip(idx[-4:-1], idx[-3:]):
    t0=int(rows[a]["Start_Timestamp"])
    busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows[a:b])
    print("step (rollout -> next rollout): %.1f us, kernels %.1f us, gaps %.1f us" % ((int(rows[b]["Start_Timestamp"])-t0)/1e3, busy/1e3, (int(rows[b]["Start_Timestamp"])-t0-busy)/1e3))
a,b=idx[-2],idx[-1]
t0=int(rows[a]["Start_Timestamp"])
upd=[i for i in range(a,b) if "ppo_update_split" in rows[i]["Kernel_Name"]]
for i in range(a,b+1):
    r=rows[i]; k=short(r["Kernel_Name"])
    if upd and upd[1] <= i <= upd[-1]+1 and ("ppo_update_split" in k or k.startswith("adam")): continue
    print("  t=%7.1f  %-36s %7.2f us  gap in front %5.2f" % ((int(r["Start_Timestamp"])-t0)/1e3, k, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, (int(r["Start_Timestamp"])-int(rows[i-1]["End_Timestamp"]))/1e3))
if upd:
    import statistics as st
    g1=[(int(rows[i]["Start_Timestamp"])-int(rows[i-1]["End_Timestamp"]))/1e3 for i in upd]
    g2=[(int(rows[i+1]["Start_Timestamp"])-int(rows[i]["End_Timestamp"]))/1e3 for i in upd]
    print("  18 x (update %.2f us, gap %.2f; adam %.2f us, gap %.2f)" % (st.mean((int(rows[i]["End_Timestamp"])-int(rows[i]["Start_Timestamp"]))/1e3 for i in upd), st.mean(g1), st.mean((int(rows[i+1]["End_Timestamp"])-int(rows[i+1]["Start_Timestamp"]))/1e3 for i in upd), st.mean(g2)))
PY
rm -rf gpurun_out/tl
