#!/usr/bin/env bash
# (GPU box, from the repo root) rocprofv3 kernel trace of a few steps of a bench workload -> the kernels of ONE step (rollout ->
# next rollout), aggregated by name: count, total and per-launch duration.      bash tools/job_step_kernels.sh <default|c3ppo|c3|tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
wl=${1:-default}
args="--steps 8 --warmup 4"; [ $wl != default ] && args="--workload $wl $args"; [ $wl = default ] && args="--no-cpu-baseline --no-c3-grid --no-batch64 $args"
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py $args > /dev/null 2> gpurun_out/tl.err
python - <<'PY'
import csv, glob, re, collections
rows=[]
for f in glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def short(n):
    m=re.search(r"(\w+_kernel(?:<[^>(]*>)?)",n); return m.group(1) if m else n[:50]
idx=[i for i,r in enumerate(rows) if "rollout_" in r["Kernel_Name"]]
a,b=idx[-2],idx[-1]
t0=int(rows[a]["Start_Timestamp"])
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows[a:b])
print("step (rollout -> next rollout): %.1f us, kernels %.1f us, %d launches (under the profiler)" % ((int(rows[b]["Start_Timestamp"])-t0)/1e3, busy/1e3, b-a))
agg=collections.OrderedDict()
for i in range(a,b):
    r=rows[i]; k=short(r["Kernel_Name"])
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    agg.setdefault(k,[0,0.0]); agg[k][0]+=1; agg[k][1]+=d
for k,(n,d) in sorted(agg.items(), key=lambda x:-x[1][1]):
    print("  %-56s x%-4d %9.2f us total %8.2f us each" % (k,n,d,d/n))
PY
rm -rf gpurun_out/tl
