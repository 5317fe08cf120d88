#!/usr/bin/env python3
"""Per-step wall time on the GPU timeline (one HIP event per step) and on the host: where does step-time jitter come from?

    python tools/step_jitter.py [n_steps]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class A:
    n_env, n_agent, horizon, minibatch, repeat, dispatch = 1024, 3, 25, 4096, 1, "per_agent"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    a = A()
    env, net, algo, buf, col = bench.build_job(a, torch.device("cuda"), 0)
    for _ in range(20):
        bench.one_step(a, algo, buf, col)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = np.zeros(n + 1)
    ev[0].record()
    host[0] = time.perf_counter()
    mode = sys.argv[2] if len(sys.argv) > 2 else "free"
    prev = None
    for i in range(n):
        if mode.startswith("lib"):  # the library alone: nobody reads any statistics (run-ahead bounded by its rings)
            from tianshou_marl_amd.algorithm.ppo import policy_within_training_step
            with policy_within_training_step(algo):
                cs = col.collect(n_step=a.n_env * a.horizon)
                ts = algo.update(buf, a.minibatch, a.repeat)
            col.reset_buffer(keep_statistics=True)
            if mode == "lib1" and prev is not None:
                prev[2].synchronize()
            e = torch.cuda.Event(); e.record()
            prev = (cs, ts, e)
            ev[i + 1].record()
            host[i + 1] = time.perf_counter()
            continue
        cs, ts = bench.one_step(a, algo, buf, col)
        if mode == "resolve":      # read this step's collect statistics (waits for the rollout, not for the update)
            cs.resolve()
        elif mode == "resolve_prev":  # read the previous step's statistics of both
            if prev is not None:
                prev[0].resolve(); prev[1].resolve()
            prev = (cs, ts)
        ev[i + 1].record()
        host[i + 1] = time.perf_counter()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    gpu = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]) * 1e3
    hst = np.diff(host) * 1e6
    pct = lambda x: np.round(np.percentile(x, [0, 10, 50, 90, 99, 100]), 0)  # noqa: E731
    print("wall per step (us): %.1f" % ((t_end - host[0]) / n * 1e6))
    print("gpu-timeline step us  min/p10/p50/p90/p99/max:", pct(gpu))
    print("host enqueue  step us min/p10/p50/p90/p99/max:", pct(hst))
    slow = np.argsort(gpu)[-8:]
    print("slowest steps (index, gpu us, host us):", [(int(i), int(gpu[i]), int(hst[i])) for i in sorted(slow)])
    # how far the host runs ahead of the device: host time of enqueue vs device completion is not observable
    # directly; a host step much shorter than the GPU step means the host is queueing ahead (GPU-bound)


if __name__ == "__main__":
    main()
