cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --workload c3ppo --steps 5 --warmup 3 > /dev/null 2> gpurun_out/tl.err
python tools/step_timeline.py gpurun_out/tl actor_rows64 6
rm -rf gpurun_out/tl
