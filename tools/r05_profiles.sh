#!/usr/bin/env bash
# Round-5 profiles on a GPU box (run through gpurun from the repo root): the default bench line, rocprofv3 kernel stats of the same
# command (md + json: the bench reads the in-situ averages from the json), the PMC passes (FETCH_SIZE / WRITE_SIZE in separate
# runs, as the guide prescribes; issue-slot counters in a third), the other workloads' lines and the kernels of one job step.
# Outputs land in gpurun_out/r05_*; copy the summaries into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_issue
timeout -k 10 400 python bench.py > $O/r05_bench.json 2> $O/r05_bench.err
echo plain done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --no-cpu-baseline --no-batch64 --steps 20 --warmup 5 > $O/r05_bench_under_rocprof.json 2> $O/r05_bench_rocprof.err
python tools/summarize_profile.py $O/prof $O/r05_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-batch64 --steps 20 --warmup 5" > /dev/null
rm -rf $O/prof
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --no-batch64 --steps 5 --warmup 3 > /dev/null 2> $O/r05_pmc_f.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --no-cpu-baseline --no-batch64 --steps 5 --warmup 3 > /dev/null 2> $O/r05_pmc_w.err
echo write done
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/r05_pmc_traffic > /dev/null
rm -rf $O/pmc_fetch $O/pmc_write
for wl in default c3ppo; do
  args="--steps 4 --warmup 3"; [ $wl != default ] && args="--workload $wl $args"; [ $wl = default ] && args="--no-cpu-baseline --no-batch64 $args"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace \
      --output-format csv -d $O/pmc_issue -o i -- python3 bench.py $args > /dev/null 2> $O/r05_pmc_issue_$wl.err
  python tools/pmc_issue.py $O/pmc_issue $O/r05_pmc_issue_$wl.md > /dev/null
  rm -rf $O/pmc_issue
  echo issue $wl done
done
for wl in c3ppo c3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --workload $wl --steps 5 --warmup 3 > /dev/null 2> $O/r05_${wl}_rocprof.err
  python tools/summarize_profile.py $O/prof $O/r05_${wl}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $wl --steps 5 --warmup 3" > /dev/null
  rm -rf $O/prof
  python bench.py --workload $wl --steps 40 --warmup 5 > $O/r05_bench_$wl.json 2> $O/r05_bench_$wl.err
  echo $wl done
done
python bench.py --workload tag > $O/r05_bench_tag.json 2> $O/r05_bench_tag.err
python bench.py --workload tag --tag-envs 4096 > $O/r05_bench_tag_4096envs.json 2>> $O/r05_bench_tag.err
bash tools/job_step_kernels.sh tag > $O/r05_tag_step_kernels.txt 2>&1
bash tools/job_step_kernels.sh default > $O/r05_default_step_kernels.txt 2>&1
ls $O | grep r05_ | head -50
