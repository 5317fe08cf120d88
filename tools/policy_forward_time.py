#!/usr/bin/env python3
"""Duration of tsm_policy_forward (64-wide actor + critic, csrc/mlp_fused.hip) on B rows per launch-grid cap (option dbg >> 8)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import ops
from tianshou_marl_amd.utils.net import DiscreteActorCritic
dev = "cuda"
D = 16
net = DiscreteActorCritic(D, 5, 64, device=dev, seed=1)
for B in (12800, 3072, 76800, 819200):
    obs = torch.randn(B, D, device=dev)
    line = []
    for cap in (0, 128, 256, 384, 512, 768):
        ops.set_kernel_option("dbg", cap << 8)
        for _ in range(3): out = ops.policy_forward(net.flat.data, obs, 5, 64, image=None, mode="none", want_logits=False)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): out = ops.policy_forward(net.flat.data, obs, 5, 64, image=None, mode="none", want_logits=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); e0.record(); [g.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
        line.append("cap %d: %.2f us" % (cap, e0.elapsed_time(e1) * 1e3 / 100))
    ops.set_kernel_option("dbg", 0)
    print(B, "rows:", ", ".join(line))
