#!/usr/bin/env python3
"""Duration of one 25-step collect of the 64-wide persistent rollout (csrc/rollout.hip) per kernel form and problem size
(HIP events around 20 collects; includes the host's launch gaps, equal for both forms).

    python tools/time_rollout64.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import ops  # noqa: E402
from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step  # noqa: E402
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer  # noqa: E402
from tianshou_marl_amd.data.collector import Collector  # noqa: E402
from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv  # noqa: E402
from tianshou_marl_amd.utils.host import limit_host_threads  # noqa: E402
from tianshou_marl_amd.utils.net import DiscreteActorCritic  # noqa: E402

limit_host_threads()
dev, T = "cuda", 25
for E, N in [(512, 3), (1024, 3), (1536, 3), (2048, 3), (4096, 3), (1024, 8), (4096, 8)]:
    env = DeviceSimpleSpreadVectorEnv(E, N, max_cycles=T, device=dev, seed=1)
    net = DiscreteActorCritic(env.obs_dim, env.n_act, 64, device=dev, seed=1)
    algo = PPO(net=net, seed=1)
    buf = DeviceVectorReplayBuffer(E * T, E, N, env.obs_dim, device=dev)
    col = Collector(algo, env, buf)
    col.reset()
    res = []
    for form in (1, 2):
        ops.set_kernel_option("rollout_form", form)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        with policy_within_training_step(algo):
            for i in range(23):
                if i == 3:
                    ev[0].record()
                col.collect(n_step=E * T)
                col.reset_buffer(keep_statistics=True)
            ev[1].record()
        torch.cuda.synchronize()
        res.append(ev[0].elapsed_time(ev[1]) * 1e3 / 20)
    waves = -(-E // (16 // N))
    print(f"{E:5d} envs x {N} agents ({waves:4d} waves): tile form {res[0]:7.1f} us, wave form {res[1]:7.1f} us per collect + reset_buffer")
ops.set_kernel_option("rollout_form", 0)
