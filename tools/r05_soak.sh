#!/usr/bin/env bash
# Round-5 soak / determinism record (run through gpurun from the repo root): every job twice, the two hash lines must agree.
# Jobs: the paths this round changed -- the headline job (update kernel: prologue in two round trips, per-layer slab stores, the critic's
# phases read ahead; rollout: pair-force tasks, index algebra beside the head, uniforms drawn ahead), the tag job (the same rollout
# changes, learners on the stored rollout outputs), the C3 PPO job (actor kernel: operands read ahead, ReLU bits, LDS-buffered rows)
# with and without a gradient-norm clip, and the CTDE job.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_soak_determinism.txt
echo "# tools/soak_determinism.py, two independent runs per job (MI355X, round 5): equal sha256 lines = bit-reproducible" > $out
for run in 1 2; do
  echo "## C3 PPO job (gradient-norm clip: reduce_slabs_segs + adam_step), 600 steps, run $run" >> $out
  timeout -k 10 300 python tools/soak_determinism.py 600 --c3 2>/dev/null | awk 'NR % 3 == 1 || /sha256/' >> $out
  echo "## C3 PPO job (no clip: one segmented Adam launch; async statistics), 600 steps, run $run" >> $out
  timeout -k 10 300 python tools/soak_determinism.py 600 --c3 --noclip 2>/dev/null | awk 'NR % 3 == 1 || /sha256/' >> $out
  echo "## CTDE job (8 learn() calls per step on the stores, one graph replay each), 300 steps, run $run" >> $out
  timeout -k 10 300 python tools/soak_determinism.py 300 --ctde 2>/dev/null | awk 'NR % 2 == 1 || /sha256/' >> $out
  echo "## headline job, gamma 0.95 / max_grad_norm 0.5, 5000 steps, run $run" >> $out
  timeout -k 10 300 python tools/soak_determinism.py 5000 --stable 2>/dev/null | awk 'NR % 4 == 1 || /sha256/' >> $out
  echo "## tag job, 2000 steps, run $run" >> $out
  timeout -k 10 300 python tools/soak_determinism.py 2000 --tag 2>/dev/null | awk 'NR % 4 == 1 || /sha256/' >> $out
done
grep sha256 $out
