#!/usr/bin/env python3
"""Phase time stamps of the second-generation critic step, kernel A (csrc/critic_train.hip), workgroup 0, 100 MHz wall clock.

    python tools/stamp_critic_train.py            # PPO value term, 8192 joint rows of 384 (one tile per workgroup)
    python tools/stamp_critic_train.py img        # the same PPO step with the first-layer weights from the fragment image
    python tools/stamp_critic_train.py td         # CTDE TD loss, 102 400 rows (13 tiles per workgroup): tiles 0, 1, 2
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi, ops  # noqa: E402
from tianshou_marl_amd.utils.net import FlatMLP  # noqa: E402

dev = "cuda"
td = len(sys.argv) > 1 and sys.argv[1] == "td"
use_img = len(sys.argv) > 1 and sys.argv[1] == "img"
D, N, H = 48, 8, 128
torch.manual_seed(0)
st = torch.zeros(1024, dtype=torch.int64, device=dev)
lib = _abi.load()
lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]
if td:
    T, E, n_out = 25, 4096, 8
    f = FlatMLP([N * D, H, H, n_out], device=dev, seed=1)
    joint, rew = torch.randn(T, E, N * D, device=dev), torch.randn(T, E, N, device=dev)
    term = torch.zeros(T, E, N, dtype=torch.uint8, device=dev)
    v_last = torch.randn(E, device=dev)
    ws: dict = {}
    run = lambda: ops.critic_rows_grad_td(f.flat.data, joint, T, E, rew, term, 3, N, v_last, 0.99, n_out, H, ws=ws)  # noqa: E731
else:
    rows, Mr = 102400, 8192
    f = FlatMLP([N * D, H, H, 1], device=dev, seed=1)
    joint, ret = torch.randn(rows, N * D, device=dev), torch.randn(rows * N, device=dev)
    rid = torch.randperm(rows, device=dev)[:Mr].contiguous()
    cfg = ops.make_ppo_cfg(value_group=N)
    ws = {}
    img = ops.critic_w1_image(f.flat.data, N * D) if use_img else None
    run = lambda: ops.critic_rows_grad_ppo(f.flat.data, joint, ret, cfg, N, H, rows=rid, Mr=Mr, ws=ws, w1_image=img)  # noqa: E731
for _ in range(3):
    run()
lib.tsm_debug_set_stamps(st.data_ptr())
for _ in range(1 if os.environ.get("STAMP_COLD") == "1" else 1500):   # (the stamps kept are the last launch's: warm clocks, not a cold one)
    run()
torch.cuda.synchronize()
lib.tsm_debug_set_stamps(None)
s = st.cpu().numpy()
names = ["L1 (W1 in registers, X by ds_read_b128) -> H1", "commit next X + fetch + L2 -> H2 (PPO: + V partials)",
         "L3 (MFMA, 2 waves; TD only) -> Q", "loss head", "dW3 / db3 / dH2", "dW2 + dH1 -> global"]
print("PPO value term" + (", W1 from the fragment image" if use_img else ", W1 gathered from the flat vector") if not td else "CTDE TD loss")
print(f"prologue (W1 fragments, W2 / W3 staging, row ids -> first tile in LDS) {(s[300] - s[298]) / 100.0:.2f} us")
for it in range(3 if td else 1):
    b = 300 + 16 * it
    d = [(s[b + k + 1] - s[b + k]) / 100.0 for k in range(6)]
    print(f"tile {it}: total {(s[b + 6] - s[b]) / 100.0:.2f} us   " + ", ".join(f"{n_} {x:.2f}" for n_, x in zip(names, d)))
print(f"slab + statistics after the last tile: {(s[299] - s[300 + 16 * (2 if td else 0) + 6]) / 100.0:.2f} us"
      + ("  (tiles 3.. of this workgroup in between)" if td else "") + f";  kernel entry -> exit of workgroup 0: {(s[299] - s[298]) / 100.0:.2f} us")
