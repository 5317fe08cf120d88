# Same-box A/B of two builds of the library for the tag job (gpurun from the repo root): tianshou_marl_amd/lib/base.so against
# lib/new.so -- tag parity tests on the new build first, then the tag bench line alternating new / base twice, then the rollout's stamps.
set -e
O=gpurun_out/abtag; mkdir -p $O
L=tianshou_marl_amd/lib
cp $L/new.so $L/libtsmarl_hip.so
timeout -k 10 500 python -m pytest tests/test_gpu_tag.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do
  cp $L/new.so $L/libtsmarl_hip.so
  timeout -k 10 200 python bench.py --workload tag > $O/new_$i.json 2> $O/new_$i.err
  cp $L/base.so $L/libtsmarl_hip.so
  timeout -k 10 200 python bench.py --workload tag > $O/base_$i.json 2> $O/base_$i.err
done
cp $L/new.so $L/libtsmarl_hip.so
timeout -k 10 200 python tools/stamp_rollout_tag.py > $O/stamp_new.txt 2>&1 || true
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/abtag/*.json')):
    d=json.load(open(f)); print(f, round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['collect_ms'],4))
P
