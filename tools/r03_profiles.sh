#!/usr/bin/env bash
# Round-3 profiles on a GPU box (run through gpurun from the repo root): bench lines, rocprofv3 kernel stats, and the PMC
# passes (FETCH_SIZE / WRITE_SIZE in separate runs, as the guide prescribes).  Outputs land in gpurun_out/r03_*; copy the
# summaries into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
timeout -k 10 500 python bench.py > $O/r03_bench.json 2> $O/r03_bench.err
echo plain done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/r03_bench_under_rocprof.json 2> $O/r03_bench_rocprof.err
python tools/summarize_profile.py $O/prof $O/r03_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5" > /dev/null
rm -rf $O/prof
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2> $O/r03_pmc_f.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2> $O/r03_pmc_w.err
echo write done
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/r03_pmc_traffic > /dev/null
rm -rf $O/pmc_fetch $O/pmc_write
for wl in c3ppo c3; do
  python bench.py --workload $wl --steps 40 --warmup 5 > $O/r03_bench_$wl.json 2> $O/r03_bench_$wl.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --workload $wl --steps 5 --warmup 3 > /dev/null 2> $O/r03_${wl}_rocprof.err
  python tools/summarize_profile.py $O/prof $O/r03_${wl}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $wl --steps 5 --warmup 3" > /dev/null
  rm -rf $O/prof
  echo $wl done
done
# the rollout's HBM writes with and without an obs_next store (C3 collect: 4096 envs x 8 agents x 25 steps)
for v in "" "--ignore-obs-next"; do
  tag=$([ -z "$v" ] && echo full || echo ignore_obs_next)
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --workload c3ppo $v --steps 4 --warmup 3 > /dev/null 2> $O/r03_pmc_w2.err
  python - <<PY
import csv, glob
rows = [r for f in glob.glob("$O/pmc_write/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
w = [float(r["Counter_Value"]) for r in rows if "rollout_rows_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
open("$O/r03_rollout_write_size.txt", "a").write("rollout_rows_kernel<3>, buffer %s: WRITE_SIZE %.1f MB per collect (mean of %d launches; KiB units of the counter x 1024)\n" % ("$tag", sum(w) / len(w) * 1024 / 1e6, len(w)))
PY
  rm -rf $O/pmc_write
done
python bench.py --workload c3ppo --ignore-obs-next --steps 40 --warmup 5 > $O/r03_bench_c3ppo_ignore_obs_next.json 2> /dev/null
python tools/c3_step_times.py 80 > $O/r03_c3_step_times.txt 2>&1
python tools/stamp_critic_train.py > $O/r03_stamp_critic_train.txt 2>&1
python tools/stamp_critic_train.py td >> $O/r03_stamp_critic_train.txt 2>&1
ls $O | grep r03_ | head -50
