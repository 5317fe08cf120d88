#!/usr/bin/env bash
# The one-off 30-60 ms hole in an early step of a process ("29 ms update"): what it is, and that capping
# torch's thread pool to the CPU quota removes it.  Run through gpurun from the repo root; writes gpurun_out/r03_host_stall.txt.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_host_stall.txt
{
echo "# tools/host_stall_probe.sh on an MI355X box.  cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max)  (quota / period in us), nproc: $(nproc)"
python -c "import torch; print('# torch default threads:', torch.get_num_threads())" 2>/dev/null
echo "# (1) torch's default pool, 8 processes x 60 steps of the C3 PPO loop without per-step synchronisation (the bench's loop):"
for i in 1 2 3 4 5 6 7 8; do python tools/c3_step_times.py 60 nosync nolimit 2>/dev/null | grep "^step  *[0-9][0-9]\|^step  *[6-9]\|^hipGraph\|^cgroup"; done
echo "# (2) the pool capped to the quota (limit_host_threads), 8 processes x 60 steps:"
for i in 1 2 3 4 5 6 7 8; do python tools/c3_step_times.py 60 nosync 2>/dev/null | grep "^step  *[0-9][0-9]\|^step  *[6-9]\|^hipGraph\|^cgroup\|^host"; done
echo "# (3) where the hole sits on the device timeline: rocprofv3 --kernel-trace of default-pool runs, largest gaps between consecutive kernels"
echo "#     (gaps in the first ~350 ms belong to the eager first update and the capture; a gap BETWEEN TWO KERNEL NODES OF THE UPDATE GRAPH is the stall)"
for i in 1 2 3; do rm -rf gpurun_out/gap; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -o g -- python3 tools/c3_step_times.py 60 nosync nolimit 2>/dev/null | grep "^step  *[0-9][0-9]\|^step  *[6-9]"; python tools/kernel_gaps.py gpurun_out/gap | head -5; rm -rf gpurun_out/gap; done
echo "# (4) bench.py --workload c3ppo --steps 40 --warmup 5, four processes (pool capped by bench.py): ms_per_step, env-steps/s"
for i in 1 2 3 4; do python bench.py --workload c3ppo --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3ppo', round(d['ms_per_step'],3), round(d['value']/1e6,1))"; done
} > $O 2>&1
cat $O
