#!/usr/bin/env bash
# (GPU box, from the repo root) rocprofv3 kernel trace of a few steps of the C3 PPO job -> the in-situ timeline of its gradient
# steps (tools/step_timeline.py) and of one whole step.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --workload c3ppo --steps 5 --warmup 3 > /dev/null 2> gpurun_out/tl.err
python tools/step_timeline.py gpurun_out/tl actor_rows64 6 | tail -8
python - <<'PY'
import csv, glob, re
rows=[]
for f in glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def short(n):
    m=re.search(r"(\w+_kernel)",n); return m.group(1) if m else n[:50]
idx=[i for i,r in enumerate(rows) if "rollout_" in r["Kernel_Name"]]
a,b=idx[-2],idx[-1]
t0=int(rows[a]["Start_Timestamp"])
print("one whole step (rollout -> next rollout): %.1f us" % ((int(rows[b]["Start_Timestamp"])-t0)/1e3))
seen_actor=0
for i in range(a,b):
    r=rows[i]; k=short(r["Kernel_Name"])
    if "actor_rows64" in k: seen_actor+=1
    if seen_actor>1 and any(x in k for x in ("actor_rows64","critic_rows_train","critic_dw1","adam_segs")): continue
    print("  t=%8.1f  %-40s %8.2f us  gap %5.2f" % ((int(r["Start_Timestamp"])-t0)/1e3, k, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, (int(r["Start_Timestamp"])-int(rows[i-1]["End_Timestamp"]))/1e3))
PY
rm -rf gpurun_out/tl
