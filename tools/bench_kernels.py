#!/usr/bin/env python3
"""Kernel micro-benchmarks (roofline grid of SURVEY.md 8d): GAE and PPO-loss kernels.

    python tools/bench_kernels.py [--iters 50]
Prints one JSON line per case: algorithmic bytes / measured kernel time vs the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import ops  # noqa: E402

PEAK = 8.0e12


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = "cuda"
    for (n_env, N, T) in [(1024, 3, 25), (4096, 8, 25), (4096, 8, 256), (4096, 8, 2048)]:
        L = n_env * N
        v_s, v_n, rew = (torch.randn(T, L, device=dev) for _ in range(3))
        te = (torch.rand(T, L, device=dev) < 0.01).to(torch.uint8)
        tr = torch.zeros(T, L, dtype=torch.uint8, device=dev)
        out = (torch.empty(T, L, device=dev), torch.empty(T, L, device=dev))
        sec = timeit(lambda: ops.gae_lanes(v_s, v_n, rew, te, tr, out=out), a.iters)
        byts = 22 * T * L
        print(json.dumps(dict(kernel="gae_lanes", n_env=n_env, n_agent=N, T=T, us=sec * 1e6,
                              GBps=byts / sec / 1e9, frac=byts / sec / PEAK, bytes=byts)))
        del v_s, v_n, rew, te, tr, out
    cfg = ops.make_ppo_cfg()
    for M in (4096, 76800, 819200, 8192000):
        A = 5
        logits = torch.randn(M, A, device=dev)
        value, logp_old, adv, ret = (torch.randn(M, device=dev) for _ in range(4))
        act = torch.randint(0, A, (M,), dtype=torch.int32, device=dev)
        stats = ops.ppo_adv_stats(adv, torch.tensor([0, M], device=dev))
        sec = timeit(lambda: ops.ppo_loss_fwd_bwd(logits, value, act, logp_old, adv, ret, cfg, adv_stats=stats[0]), a.iters)
        byts = 64 * M
        print(json.dumps(dict(kernel="ppo_loss_fwd_bwd(+finalize,+allocs)", M=M, us=sec * 1e6, GBps=byts / sec / 1e9,
                              frac=byts / sec / PEAK, bytes=byts)))


if __name__ == "__main__":
    main()
