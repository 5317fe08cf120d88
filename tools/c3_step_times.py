#!/usr/bin/env python3
"""Per-step timeline of the C3 PPO job (4096 envs x 8 agents, GenericPPO with a centralized critic): for every step the
host wall time and, from HIP events on the launch stream, the device time of collect and of update -- which steps are
slow, and whether the time is on the device (inside the bracket) or on the host (outside it).

    python tools/c3_step_times.py [n_steps] [nosync] [nolimit]

`nosync`: no per-step synchronisation (the bench's own loop); the events are read after the last step.

Runs once; prints one line per step that is slower than 1.3x the median, and the labels of the first steps (eager update,
capture, first replay)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd.algorithm import GenericPPO, policy_within_training_step  # noqa: E402
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer  # noqa: E402
from tianshou_marl_amd.data.collector import Collector  # noqa: E402
from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv  # noqa: E402
from tianshou_marl_amd.utils.net import MLPActorCritic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    nosync = "nosync" in sys.argv[2:]
    if "nolimit" not in sys.argv[2:]:  # `nolimit`: torch's default pool (128 threads on these hosts) -- reproduces the stall
        from tianshou_marl_amd.utils.host import cpu_budget, limit_host_threads

        print("host threads: torch pool %d of a CPU budget of %d" % (limit_host_threads(), cpu_budget()))
    dev = torch.device("cuda", 0)
    n_env, N, T, mb = 4096, 8, 25, 65536
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=dev, seed=1626)
    D = env.obs_dim
    net = MLPActorCritic(D, 5, (128, 128), critic_obs_dim=N * D, device=dev, seed=1626)
    algo = GenericPPO(net=net, critic_input="global", n_agent=N, lr=3e-4, shuffle="device", seed=1626, dispatch="pooled")
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, D, device=dev)
    col = Collector(algo, env, buf, async_stats=True)
    col.reset()
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    rec = []
    import gc

    gc_log, gc_t = [], [0.0]

    def on_gc(phase, info):  # how long the interpreter's cyclic collector holds the loop, and in which step
        if phase == "start":
            gc_t[0] = time.perf_counter()
        else:
            gc_log.append((len(rec), info["generation"], (time.perf_counter() - gc_t[0]) * 1e3, info["collected"]))

    gc.callbacks.append(on_gc)
    # time every hipGraph launch on the host: wall, and the calling thread's CPU time (wall >> cpu: the thread slept or was
    # descheduled inside hipGraphLaunch; wall == cpu: it was working or spinning)
    replays = []
    _replay = torch.cuda.CUDAGraph.replay

    def timed_replay(self):
        w0, c0 = time.perf_counter(), time.thread_time()
        _replay(self)
        replays.append((len(rec), (time.perf_counter() - w0) * 1e3, (time.thread_time() - c0) * 1e3))

    torch.cuda.CUDAGraph.replay = timed_replay

    def cpu_stat():
        try:
            return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
        except OSError:
            return {}

    cs0 = cpu_stat()
    for i in range(n):
        e = [ev() for _ in range(3)]
        t0 = time.perf_counter()
        with policy_within_training_step(algo):
            e[0].record()
            cs = col.collect(n_step=n_env * T)
            t1 = time.perf_counter()
            e[1].record()
            algo.update(buf, mb, 1)
            t2 = time.perf_counter()
            e[2].record()
        r = getattr(cs, "resolve", None)
        if callable(r):
            r()
        t3 = time.perf_counter()
        col.reset_buffer(keep_statistics=True)
        t4 = time.perf_counter()
        if nosync:
            rec.append([t4 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3, e])
            continue
        torch.cuda.synchronize()
        rec.append((t4 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
    if nosync:  # the events are read once, after the loop
        torch.cuda.synchronize()
        rec = [(*r[:5], r[5][0].elapsed_time(r[5][1]), r[5][1].elapsed_time(r[5][2])) for r in rec]
    a = np.array(rec) * np.array([1e3, 1e3, 1e3, 1e3, 1e3, 1, 1])
    med = np.median(a[5:], axis=0)
    names = "wall host_collect host_update host_resolve host_reset dev_collect dev_update".split()
    print("median over steps 5.. (ms): " + "  ".join(f"{k} {v:.3f}" for k, v in zip(names, med)))
    label = {0: "eager update (first of its shape)", 1: "update captured into the hipGraph (+ its first replay)", 2: "replay"}
    for i, row in enumerate(a):
        if i < 6 or row[0] > 1.3 * med[0]:
            print(f"step {i:3d} " + "  ".join(f"{k} {v:.3f}" for k, v in zip(names, row)) + ("   <- " + label[i] if i in label else ""))
    cs1 = cpu_stat()
    print("cgroup cpu.stat over the loop: " + ", ".join(f"{k} +{cs1[k] - cs0[k]}" for k in cs1 if k in cs0 and ("throttled" in k or k == "nr_periods")))
    for st, wall, cpu in replays:
        if wall > 1.0:
            print(f"hipGraphLaunch in step {st}: {wall:.2f} ms on the host, {cpu:.2f} ms of them on the CPU")
    for st, gen, ms, coll in gc_log:
        if ms > 1.0:
            print(f"python gc: generation {gen} collection during step {st}: {ms:.1f} ms ({coll} objects freed)")
    print("steps over 1.3x the median wall: %d of %d" % (int((a[:, 0] > 1.3 * med[0]).sum()), n))


if __name__ == "__main__":
    main()
