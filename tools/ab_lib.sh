# Same-box A/B of two builds of the library (gpurun from the repo root): tianshou_marl_amd/lib/base.so (e.g. built from `git stash`)
# against lib/new.so -- rollout parity tests on the new build first, then the default bench line alternating new / base twice.
set -e
O=gpurun_out/ab1; mkdir -p $O
L=tianshou_marl_amd/lib
timeout -k 10 400 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_marl.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do
  cp $L/new.so $L/libtsmarl_hip.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-batch64 > $O/new_$i.json 2> $O/new_$i.err
  cp $L/base.so $L/libtsmarl_hip.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-batch64 > $O/base_$i.json 2> $O/base_$i.err
done
cp $L/new.so $L/libtsmarl_hip.so
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab1/*.json')):
    d=json.load(open(f)); print(f, round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['collect_ms'],4), round(d['ppo_update_ms'],4))
P
