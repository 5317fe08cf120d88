set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_dense.py tests/test_gpu_generic_ppo.py -x -q -m gpu 2>&1 | tail -2
rm -rf $O/prof
for wl in c3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --workload $wl --steps 5 --warmup 3 > /dev/null 2> $O/r04_${wl}_rocprof.err
  python tools/summarize_profile.py $O/prof $O/r04_${wl}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $wl --steps 5 --warmup 3" > /dev/null
  rm -rf $O/prof
  python bench.py --workload $wl --steps 40 --warmup 5 > $O/r04_bench_$wl.json 2> $O/r04_bench_$wl.err
done
python -c "
import json;d=json.load(open('gpurun_out/r04_bench_c3.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'])"
head -14 $O/r04_c3_kernel_stats.md | tail -6
