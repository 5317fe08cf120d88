#!/usr/bin/env bash
# (GPU box, from the repo root) issue-slot accounting of the bench workloads -> gpurun_out/r04_pmc_issue_<workload>.{md,json}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for wl in default c3ppo c3; do
  rm -rf $O/pmc_issue
  args="--steps 4 --warmup 3"; [ $wl != default ] && args="--workload $wl $args"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace \
      --output-format csv -d $O/pmc_issue -o i -- python3 bench.py $args > /dev/null 2> $O/pmc_issue_$wl.err
  python tools/pmc_issue.py $O/pmc_issue $O/r04_pmc_issue_$wl.md | head -12
  rm -rf $O/pmc_issue
done
