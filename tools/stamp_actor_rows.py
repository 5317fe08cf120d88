#!/usr/bin/env python3
"""Phase time stamps of the one-launch actor step at BASELINE configs[2]: 65 536 samples of 48 -> 128 -> 128 -> 5, workgroup 0,
its tiles 1-3 (100 MHz wall clock).  The minibatch size picks the kernel (csrc/actor_rows64.hip: 64-sample tiles, W2 in
registers; csrc/ppo_rows.hip: 32-sample tiles); TSM_ACTOR_TILE=32 | 64 forces one.

    python tools/stamp_actor_rows.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi, ops  # noqa: E402
from tianshou_marl_amd.utils.net import MLPActorCritic  # noqa: E402

dev = "cuda"
D, A, H, n, M = 48, 5, 128, 819200, 65536
torch.manual_seed(0)
net = MLPActorCritic(D, A, (H, H), critic_obs_dim=8 * D, device=dev, seed=1)
obs = torch.randn(n, D, device=dev)
act = torch.randint(0, A, (n,), dtype=torch.int32, device=dev)
lp, adv = torch.randn(n, device=dev) * 0.3 - 1.5, torch.randn(n, device=dev)
perm = torch.randperm(n, device=dev)[:M].contiguous()
stats = ops.ppo_adv_stats(adv, torch.tensor([0, M], device=dev), perm=perm, max_rows=M)
cfg = ops.make_ppo_cfg()
nb = ops.ppo_actor_rows_grid(M)
slabs = torch.empty(nb, net.n_actor, device=dev)
part = torch.empty(nb * 4, dtype=torch.float64, device=dev)
st = torch.zeros(1024, dtype=torch.int64, device=dev)
lib = _abi.load()
lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.ppo_actor_rows_update(net.actor.flat.data, obs, act, lp, adv, cfg, A, H, adv_stats=stats[0], perm=perm, M=M, n_blocks=nb,
                              slabs=slabs, partial=part)
lib.tsm_debug_set_stamps(st.data_ptr())
ops.ppo_actor_rows_update(net.actor.flat.data, obs, act, lp, adv, cfg, A, H, adv_stats=stats[0], perm=perm, M=M, n_blocks=nb,
                          slabs=slabs, partial=part)
torch.cuda.synchronize()
lib.tsm_debug_set_stamps(None)
sa = st.cpu().numpy()
if sa[201] > sa[200] > 0:  # (the 64-sample kernel stamps its prologue and epilogue too)
    print(f"prologue (weights -> LDS / registers, first ids and rows) {(sa[201] - sa[200]) / 100.0:.2f} us; slab + statistics after the last tile "
          f"{(sa[203] - sa[202]) / 100.0:.2f} us; kernel entry -> exit of workgroup 0 {(sa[203] - sa[200]) / 100.0:.2f} us")
s = sa[:64].reshape(4, 16)
names = ["P0 commit X", "P1 layer 1", "P2 layer 2", "P3 logits", "P4 loss head (16 lanes per sample)", "P5 dW3 + dH2 mfma", "dH2 write",
         "P6 dW2 + dH1 mfma", "dH1 write", "P7 dW1"]
for it in range(1, 4):
    d = [(s[it][k + 1] - s[it][k]) / 100.0 for k in range(10)]
    print(f"tile {it}: total {(s[it][10] - s[it][0]) / 100.0:.2f} us   " + ", ".join(f"{n_} {x:.2f}" for n_, x in zip(names, d)))
rows = 64 if -(-M // 64) >= ops.device_info()["n_cu"] else 32  # (the rule of tsm_ppo_actor_rows_grid)
if os.environ.get("TSM_ACTOR_TILE") in ("32", "64"):
    rows = int(os.environ["TSM_ACTOR_TILE"])
print(f"n_blocks {nb}, {rows}-sample tiles, {M // rows // nb} tiles per workgroup (P3: logits on {'all eight waves, two k halves' if rows == 64 else 'two waves'})")
