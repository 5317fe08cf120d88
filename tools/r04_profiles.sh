#!/usr/bin/env bash
# Round-4 profiles on a GPU box (run through gpurun from the repo root): bench lines, rocprofv3 kernel stats (md + json: the
# bench reads the in-situ averages from the json), and the PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, as the guide
# prescribes).  Outputs land in gpurun_out/r04_*; copy the summaries into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/r04_bench_under_rocprof.json 2> $O/r04_bench_rocprof.err
python tools/summarize_profile.py $O/prof $O/r04_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5" > /dev/null
rm -rf $O/prof
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2> $O/r04_pmc_f.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2> $O/r04_pmc_w.err
echo write done
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/r04_pmc_traffic > /dev/null
rm -rf $O/pmc_fetch $O/pmc_write
for wl in c3ppo c3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --workload $wl --steps 5 --warmup 3 > /dev/null 2> $O/r04_${wl}_rocprof.err
  python tools/summarize_profile.py $O/prof $O/r04_${wl}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $wl --steps 5 --warmup 3" > /dev/null
  rm -rf $O/prof
  python bench.py --workload $wl --steps 40 --warmup 5 > $O/r04_bench_$wl.json 2> $O/r04_bench_$wl.err
  echo $wl done
done
python bench.py --workload tag --steps 40 --warmup 5 > $O/r04_bench_tag.json 2> $O/r04_bench_tag.err
python tools/stamp_critic_train.py > $O/r04_stamp_critic_train.txt 2>&1
python tools/stamp_critic_train.py img >> $O/r04_stamp_critic_train.txt 2>&1
python tools/stamp_critic_train.py td >> $O/r04_stamp_critic_train.txt 2>&1
ls $O | grep r04_ | head -50
