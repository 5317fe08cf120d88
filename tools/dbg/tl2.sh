cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --no-cpu-baseline --no-c3-grid --steps 30 --warmup 10 > /dev/null 2> gpurun_out/tl.err
python - <<'PY'
import csv, glob, re
rows=[]
for f in glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def short(n):
    m=re.search(r"(\w+_kernel)",n); return m.group(1) if m else n[:50]
idx=[i for i,r in enumerate(rows) if "rollout_kernel" in r["Kernel_Name"]]
# pick a step in the middle of the timed region
a,b=idx[25],idx[26]
t0=int(rows[a]["Start_Timestamp"])
print("one whole step (rollout -> next rollout): %.1f us; previous: %.1f, next: %.1f" % ((int(rows[b]["Start_Timestamp"])-t0)/1e3, (t0-int(rows[idx[24]]["Start_Timestamp"]))/1e3, (int(rows[idx[27]]["Start_Timestamp"])-int(rows[b]["Start_Timestamp"]))/1e3))
n_upd=0
for i in range(a,b):
    r=rows[i]; k=short(r["Kernel_Name"])
    if "ppo_update_split" in k: n_upd+=1
    if n_upd>2 and any(x in k for x in ("ppo_update_split","adam_kernel")) : continue
    print("  t=%8.1f  %-40s %8.2f us  gap %5.2f" % ((int(r["Start_Timestamp"])-t0)/1e3, k, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, (int(r["Start_Timestamp"])-int(rows[i-1]["End_Timestamp"]))/1e3))
PY
rm -rf gpurun_out/tl
