#!/usr/bin/env python3
"""Duration of one collect(n_step = 25 vector steps) of the actor-only persistent rollout at BASELINE configs[2] (4096 envs x 8
agents, actor 48-128-128-5) by HIP events, per kernel form and under the timing probes of option "dbg" (1: no matrix
products, 2: no heads / env step, 16: one wave per SIMD -- results are garbage under them).  tools/time_rollout_rows.sh runs this
under rocprofv3 and prints the kernel durations (the event times here include the host's launch gaps).

    python tools/time_rollout_rows.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import ops  # noqa: E402
from tianshou_marl_amd.algorithm import GenericPPO, policy_within_training_step  # noqa: E402
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer  # noqa: E402
from tianshou_marl_amd.data.collector import Collector  # noqa: E402
from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv  # noqa: E402
from tianshou_marl_amd.utils.host import limit_host_threads  # noqa: E402
from tianshou_marl_amd.utils.net import MLPActorCritic  # noqa: E402

limit_host_threads()
dev = "cuda"
E, N, T = 4096, 8, 25
env = DeviceSimpleSpreadVectorEnv(E, N, max_cycles=T, device=dev, seed=1)
net = MLPActorCritic(env.obs_dim, 5, (128, 128), critic_obs_dim=N * env.obs_dim, device=dev, seed=1)
algo = GenericPPO(net=net, critic_input="global", n_agent=N, seed=1, dispatch="pooled")
buf = DeviceVectorReplayBuffer(E * T, E, N, env.obs_dim, device=dev)
col = Collector(algo, env, buf)
col.reset()


def run(form: int, dbg: int, reps: int = 20) -> float:
    ops.set_kernel_option("rollout_form", form)
    ops.set_kernel_option("dbg", dbg)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with policy_within_training_step(algo):
        for i in range(reps + 3):
            if i == 3:
                ev[0].record()
            col.collect(n_step=E * T)
            col.reset_buffer(keep_statistics=True)
        ev[1].record()
    torch.cuda.synchronize()
    ops.set_kernel_option("dbg", 0)
    return ev[0].elapsed_time(ev[1]) * 1e3 / reps


for name, form, dbg in [("tile form", 1, 0), ("wave form", 2, 0), ("wave form, no matrix products", 2, 1),
                        ("wave form, no heads / env step", 2, 2), ("wave form, neither", 2, 3),
                        ("wave form, waves 0-3 only", 2, 16), ("wave form, waves 0-3 only, no env step", 2, 18),
                        ("wave form, waves 0-3 only, no products", 2, 17)]:
    us = run(form, dbg)
    print(f"{name:34s} {us:8.1f} us per collect (+ reset_buffer), {us / T:6.2f} us per vector step")
