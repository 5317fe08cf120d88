import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("tgm", "tests/test_gpu_marl.py"); tgm = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgm)
from tianshou_marl_amd.env import DummyVectorEnv, EnhancedPettingZooEnv
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
from tianshou_marl_amd.data.collector import Collector
from tianshou_marl_amd.algorithm.ppo import policy_within_training_step
n_env, N, T = 4, 3, 25
venv = DummyVectorEnv([lambda i=i: EnhancedPettingZooEnv(tgm.HostSpread(N, T, seed=i), mode="parallel") for i in range(n_env)])
algo = tgm._ppo(6 * N, 4)
buf = DeviceVectorReplayBuffer(n_env * T * 4, n_env, N, 6 * N, device="cuda")
col = Collector(algo, venv, buf); col.reset()
with policy_within_training_step(algo):
    col.collect(n_step=n_env * T)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        col.collect(n_step=n_env * T); col.reset_buffer(keep_statistics=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("host-path env-steps/s (4 envs x 3 agents):", 3 * n_env * N * T / dt)
