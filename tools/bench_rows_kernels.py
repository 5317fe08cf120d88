#!/usr/bin/env python3
"""Per-launch time of the one-launch actor / critic steps (csrc/ppo_rows.hip) at the 4096 x 8 x 25 configuration, for a
sweep of workgroup counts (= gradient slabs):  python tools/bench_rows_kernels.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import ops  # noqa: E402
from tianshou_marl_amd.utils.net import FlatMLP  # noqa: E402

DEV = "cuda"


def per_launch(fn, n=10, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3 / n / reps
    return tot


def main():
    N, D, H, A, rows_total, mb = 8, 48, 128, 5, 4096 * 25, 65536
    n = rows_total * N
    actor, critic = FlatMLP([D, H, H, A], device=DEV, seed=1), FlatMLP([N * D, H, H, 1], device=DEV, seed=2)
    obs = torch.randn(n, D, device=DEV)
    act = torch.randint(0, A, (n,), dtype=torch.int32, device=DEV)
    lp, adv, ret, v_old = (torch.randn(n, device=DEV) for _ in range(4))
    rows = torch.randperm(rows_total, device=DEV)[:mb // N].contiguous()
    perm = (rows.view(-1, 1) * N + torch.arange(N, device=DEV).view(1, -1)).reshape(-1).contiguous()
    cfg = ops.make_ppo_cfg()
    st = ops.ppo_adv_stats(adv, torch.tensor([0, mb], device=DEV), perm=perm, max_rows=mb)
    for nb in (128, 192, 256):
        us = per_launch(lambda: ops.ppo_actor_rows_update(actor.flat.data, obs, act, lp, adv, cfg, A, H, adv_stats=st[0], perm=perm,
                                                          n_blocks=nb))
        print(json.dumps(dict(kernel="actor", n_blocks=nb, us=round(us, 1), tflops=round(3 * 2 * (D * H + H * H + H * A) * mb / us / 1e6, 1))))
    joint = obs.view(rows_total, N * D)
    for nb in (64, 128, 256):
        us = per_launch(lambda: ops.ppo_critic_rows_update(critic.flat.data, joint, ret, cfg, N, H, rows=rows, n_blocks=nb))
        print(json.dumps(dict(kernel="critic", n_blocks=nb, us=round(us, 1),
                              tflops=round(3 * 2 * (N * D * H + H * H + H) * (mb // N) / us / 1e6, 1))))


if __name__ == "__main__":
    main()
