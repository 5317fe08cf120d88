for f in 0 0 0 128 0 0; do
  echo -n "dbg=$f  "
  TSM_DBG=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-c3-grid --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['collect_ms'], d['roofline']['us_per_launch'])"
done
bash tools/job_step_kernels.sh default | head -5
