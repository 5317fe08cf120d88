#!/usr/bin/env python3
"""Device-time micro-benchmarks of the per-step kernels at BASELINE config C2 sizes (graph-batched launches)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import ops  # noqa: E402
from tianshou_marl_amd.utils.net import DiscreteActorCritic  # noqa: E402

dev = "cuda"


def gtime(fn, n=20, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best


def main():
    D, A, H = 18, 5, 64
    net = DiscreteActorCritic(D, A, H, device=dev, seed=0)
    P = net.flat.data
    n = 76800
    obs = torch.randn(n, D, device=dev)
    act = torch.randint(0, A, (n,), dtype=torch.int32, device=dev)
    logp, adv, ret = (torch.randn(n, device=dev) for _ in range(3))
    cfg = ops.make_ppo_cfg()
    for M in (4096, 25600):
        perm = torch.randperm(n, device=dev)[:M].contiguous()
        stats = ops.ppo_adv_stats(adv, torch.tensor([0, M], device=dev), perm=perm)
        for nb in (256, 128, 64, 32, 16):
            if nb > ops.ppo_update_grid(M):
                continue
            slabs = torch.empty(nb, P.numel(), device=dev)
            partial = torch.empty(nb * 4, dtype=torch.float64, device=dev)
            for img in (None, net.image):
                us = gtime(lambda: ops.ppo_update_fused(P, obs, act, logp, adv, ret, cfg, A, H, adv_stats=stats[0], perm=perm,
                                                        M=M, n_blocks=nb, slabs=slabs, partial=partial, want_scalars=False,
                                                        image=img))
                print(json.dumps(dict(k="ppo_update_fused", M=M, n_blocks=nb, image=img is not None, us=round(us, 2))))
            m, v = torch.zeros_like(P), torch.zeros_like(P)
            P2 = P.clone()
            us = gtime(lambda: ops.adam_step(P2, slabs, m, v, 1))
            print(json.dumps(dict(k="adam_step", n_slab=nb, us=round(us, 2))))
    for B in (3072, 76800):
        o = obs[:B]
        out = dict(logits=None, value=torch.empty(B, device=dev), act=torch.empty(B, dtype=torch.int32, device=dev),
                   logp=torch.empty(B, device=dev))
        for img in (None, net.image):
            us = gtime(lambda: ops.policy_forward(P, o, A, H, mode="sample", seed=1, out=out, image=img))
            print(json.dumps(dict(k="policy_forward", B=B, image=img is not None, us=round(us, 2))))


if __name__ == "__main__":
    main()
