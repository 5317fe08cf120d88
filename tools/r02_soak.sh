#!/usr/bin/env bash
# Round-2 soak / determinism record (run through gpurun from the repo root): every job twice, the two hash lines must agree.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_soak_determinism.txt
echo "# tools/soak_determinism.py, two independent runs per job (MI355X, round 2): equal sha256 lines = bit-reproducible" > $out
for run in 1 2; do
  echo "## headline job, gamma 0.95 / max_grad_norm 0.5, 5000 steps, run $run" >> $out
  timeout -k 10 300 python tools/soak_determinism.py 5000 --stable 2>/dev/null | awk 'NR % 2 == 1 || /sha256/' >> $out
  echo "## C3 job, 600 steps, run $run" >> $out
  timeout -k 10 300 python tools/soak_determinism.py 600 --c3 2>/dev/null >> $out
  echo "## tag job, 3000 steps, run $run" >> $out
  timeout -k 10 300 python tools/soak_determinism.py 3000 --tag 2>/dev/null >> $out
done
grep sha256 $out
