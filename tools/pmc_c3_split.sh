#!/usr/bin/env bash
# (GPU box, from the repo root) HBM traffic of the C3 PPO job's gradient-step kernels, default vs TSM_CRITIC_SPLIT_DW2=1
# (separate --pmc FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/r04_pmc_c3ppo_split{0,1}.{md,json}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for mode in 0 1; do
  export TSM_CRITIC_SPLIT_DW2=$mode
  rm -rf $O/pf $O/pw
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o f -- python3 bench.py --workload c3ppo --steps 4 --warmup 3 > /dev/null 2> $O/pmc_s.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o w -- python3 bench.py --workload c3ppo --steps 4 --warmup 3 > /dev/null 2>> $O/pmc_s.err
  python tools/pmc_traffic.py $O/pf $O/pw $O/r04_pmc_c3ppo_split$mode > /dev/null
  echo "== TSM_CRITIC_SPLIT_DW2=$mode"
  grep -E "critic_rows_train|critic_dw1|adam_segs|actor_rows64" $O/r04_pmc_c3ppo_split$mode.md
  rm -rf $O/pf $O/pw
done
