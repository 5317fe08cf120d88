#!/usr/bin/env python3
"""Step rate of the batched simple_tag env kernel (BASELINE configs[4]: 3 adversaries + 1 good agent, 2 obstacles).

    python tools/bench_tag.py
Graph-batched back-to-back `step_device` launches timed with HIP events; prints env-steps/s (n_env x n_agent per vector
step) per n_env, next to simple_spread's stand-alone step kernel at the same sizes.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import ops  # noqa: E402
from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv  # noqa: E402
from tianshou_marl_amd.env.mpe_tag import DeviceSimpleTagVectorEnv  # noqa: E402


def time_env(env, reps=50):
    E, N = env.env_num, env.n_agent
    env.reset_device()
    act = torch.randint(0, 5, (E, N), dtype=torch.int32, device="cuda")
    for _ in range(3):
        env.step_device(act)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(reps):
            env.step_device(act)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * reps)
    return us, E * N / us * 1e6


def main():
    for n_env in (1024, 4096, 16384):
        for name, env in (("simple_tag 3v1", DeviceSimpleTagVectorEnv(n_env, device="cuda")),
                          ("simple_spread N=3", DeviceSimpleSpreadVectorEnv(n_env, 3, device="cuda"))):
            us, rate = time_env(env)
            print(json.dumps({"env": name, "n_env": n_env, "n_agent": env.n_agent, "us_per_vector_step": round(us, 2),
                              "env_steps_per_s": round(rate)}))


if __name__ == "__main__":
    main()
