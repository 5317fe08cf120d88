#!/usr/bin/env python3
"""Phase time stamps of the one-launch critic step (csrc/ppo_rows.hip, ppo_critic_rows_kernel) at BASELINE configs[2]:
8192 joint rows of 384 -> 128 -> 128 -> 1, workgroup 0 (one 32-row tile per workgroup; 100 MHz wall clock).

    python tools/stamp_critic_rows.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi, ops  # noqa: E402
from tianshou_marl_amd.utils.net import MLPActorCritic  # noqa: E402

dev = "cuda"
D, N, H, rows, Mr = 48, 8, 128, 102400, 8192
torch.manual_seed(0)
net = MLPActorCritic(D, 5, (H, H), critic_obs_dim=N * D, device=dev, seed=1)
joint = torch.randn(rows, N * D, device=dev)
ret = torch.randn(rows * N, device=dev)
rid = torch.randperm(rows, device=dev)[:Mr].contiguous()
cfg = ops.make_ppo_cfg(value_group=N)
nb = ops.ppo_critic_rows_grid(Mr)
slabs = torch.empty(nb, net.critic.flat.numel(), device=dev)
part = torch.empty(nb * 4, dtype=torch.float64, device=dev)
st = torch.zeros(1024, dtype=torch.int64, device=dev)
lib = _abi.load()
lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]
run = lambda: ops.ppo_critic_rows_update(net.critic.flat.data, joint, ret, cfg, N, H, rows=rid, Mr=Mr, n_blocks=nb, slabs=slabs,  # noqa: E731
                                         partial=part)
for _ in range(3):
    run()
lib.tsm_debug_set_stamps(st.data_ptr())
run()
torch.cuda.synchronize()
lib.tsm_debug_set_stamps(None)
s = st.cpu().numpy()[200:210]
names = ["stage W2 / W3 / biases (issue)", "layer 1 (12 K-slices of W1 + obs)", "layer 2", "value (32 lanes, 128-long fma chain)",
         "value loss of the row's agents", "dW3 / db3 / dH2", "dW2 + dH1", "dW1 (12 slices again) -> slab", "dW2 / biases -> slab"]
d = [(s[k + 1] - s[k]) / 100.0 for k in range(9)]
print(f"total {(s[9] - s[0]) / 100.0:.2f} us   " + ", ".join(f"{n_} {x:.2f}" for n_, x in zip(names, d)))
