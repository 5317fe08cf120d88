python -m pytest tests/test_gpu_generic_ppo.py tests/test_gpu_kernels.py tests/test_gpu_ppo_replay_wide.py tests/test_gpu_dense.py -x -q -m gpu > gpurun_out/r04_t3.log 2>&1; tail -3 gpurun_out/r04_t3.log
python tools/stamp_critic_train.py img 2>&1 | tail -3
python tools/dbg/pair_time.py 2>&1 | tail -3
python bench.py --workload c3ppo --steps 40 --warmup 5 > gpurun_out/r04_c3ppo_b.json 2>gpurun_out/r04_c3ppo_b.err; python -c "
import json;d=json.load(open('gpurun_out/r04_c3ppo_b.json'));print(d['value'],d['ms_per_step'],d['gae_ppo_update_ms'],d['roofline']['frac'])"
