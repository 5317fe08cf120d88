#!/usr/bin/env python3
"""Host wall time per call of the bench step's three host-side calls (is the step host- or GPU-bound?).

    python tools/host_timing.py
Runs the bench job, then times collect() / update() / reset_buffer() on the host with perf_counter, with and without
a device synchronisation after every call, and a few micro-operations (pinned D2H copy, graph replay, ctypes call).
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tianshou_marl_amd.algorithm.ppo import policy_within_training_step  # noqa: E402


class A:
    n_env, n_agent, horizon, minibatch, repeat, dispatch = 1024, 3, 25, 4096, 1, "per_agent"


def main():
    a = A()
    env, net, algo, buf, col = bench.build_job(a, torch.device("cuda"), 0)
    for _ in range(10):
        bench.one_step(a, algo, buf, col)
    torch.cuda.synchronize()
    for sync in (False, True):
        t = dict(collect=0.0, update=0.0, reset=0.0)
        n = 200
        t_all0 = time.perf_counter()
        for _ in range(n):
            with policy_within_training_step(algo):
                t0 = time.perf_counter()
                col.collect(n_step=a.n_env * a.horizon)
                if sync:
                    torch.cuda.synchronize()
                t1 = time.perf_counter()
                algo.update(buf, a.minibatch, a.repeat)
                if sync:
                    torch.cuda.synchronize()
                t2 = time.perf_counter()
            col.reset_buffer(keep_statistics=True)
            if sync:
                torch.cuda.synchronize()
            t3 = time.perf_counter()
            t["collect"] += t1 - t0
            t["update"] += t2 - t1
            t["reset"] += t3 - t2
        torch.cuda.synchronize()
        tot = time.perf_counter() - t_all0
        print(f"sync_after_each_call={sync}: " + ", ".join(f"{k} {v / n * 1e6:.0f} us" for k, v in t.items()) +
              f", loop {tot / n * 1e6:.0f} us/step")
    # micro-operations on an idle device
    x = torch.zeros(1024 * 30, dtype=torch.int64, device="cuda")
    h = torch.empty(x.shape, dtype=torch.int64, pin_memory=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        h.copy_(x, non_blocking=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"pinned D2H copy_ (240 KB) host call: {(t1 - t0) / 200 * 1e6:.1f} us")
    ev = torch.cuda.Event()
    t0 = time.perf_counter()
    for _ in range(200):
        ev.record()
    t1 = time.perf_counter()
    print(f"event.record host call: {(t1 - t0) / 200 * 1e6:.1f} us")
    t0 = time.perf_counter()
    for _ in range(200):
        net.sync_image()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"sync_image (one ctypes launch) host call: {(t1 - t0) / 200 * 1e6:.1f} us")


if __name__ == "__main__":
    main()
