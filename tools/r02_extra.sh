#!/usr/bin/env bash
# Round-2 extra measurements (run through gpurun from the repo root): slab sweep of the headline gradient step, the tag
# workload, the C3 PPO workload and its kernel stats.  Outputs land in gpurun_out/ (copy into profiles/).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/sweep_slab.py > gpurun_out/r02_sweep_slab.txt 2> gpurun_out/sweep_slab.err
echo sweep done
timeout -k 10 200 python bench.py --workload tag > gpurun_out/r02_bench_tag.json 2> gpurun_out/tag.err
echo tag done
timeout -k 10 200 python bench.py --workload c3ppo --steps 10 --warmup 4 > gpurun_out/r02_bench_c3ppo.json 2> gpurun_out/c3ppo.err
echo c3ppo done
rm -rf gpurun_out/prof_c3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -o p -- python3 bench.py --workload c3ppo --steps 5 --warmup 3 > gpurun_out/c3_rocprof.json 2> gpurun_out/c3_rocprof.err
python tools/summarize_profile.py gpurun_out/prof_c3 gpurun_out/r02_c3ppo_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload c3ppo --steps 5 --warmup 3"
rm -rf gpurun_out/prof_c3
cat gpurun_out/r02_sweep_slab.txt
