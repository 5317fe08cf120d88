timeout -k 10 400 python -m pytest tests/test_gpu_rollout.py -x -q 2>&1 | tail -2
for f in 128 0 0; do
  echo -n "c3ppo dbg=$f  "
  TSM_DBG=$f timeout -k 10 200 python bench.py --workload c3ppo --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['collect_ms'],d['gae_ppo_update_ms'])"
done
timeout -k 10 300 python tools/soak_determinism.py 600 --c3 --noclip 2>/dev/null | grep sha256
