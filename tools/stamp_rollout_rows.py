#!/usr/bin/env python3
"""Phase time stamps of the actor-only persistent rollout (csrc/rollout_rows.hip) at BASELINE configs[2]: 4096 envs x 8
agents, actor 48-128-128-5.  Workgroup 0, vector steps 1-3 (100 MHz wall clock).

    python tools/stamp_rollout_rows.py [wave|tile]      # default: the wave-autonomous form (stamps of wave 0)
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi  # noqa: E402
from tianshou_marl_amd.algorithm import GenericPPO, policy_within_training_step  # noqa: E402
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer  # noqa: E402
from tianshou_marl_amd.data.collector import Collector  # noqa: E402
from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv  # noqa: E402
from tianshou_marl_amd.utils.net import MLPActorCritic  # noqa: E402

from tianshou_marl_amd import ops  # noqa: E402

form = sys.argv[1] if len(sys.argv) > 1 else "wave"
ops.set_kernel_option("rollout_form", 1 if form == "tile" else 2)
dev = "cuda"
E, N, T = 4096, 8, 25
env = DeviceSimpleSpreadVectorEnv(E, N, max_cycles=T, device=dev, seed=1)
net = MLPActorCritic(env.obs_dim, 5, (128, 128), critic_obs_dim=N * env.obs_dim, device=dev, seed=1)
algo = GenericPPO(net=net, critic_input="global", n_agent=N, seed=1, dispatch="pooled")
buf = DeviceVectorReplayBuffer(E * T, E, N, env.obs_dim, device=dev)
col = Collector(algo, env, buf)
col.reset()
st = torch.zeros(1024, dtype=torch.int64, device=dev)
lib = _abi.load()
lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]
lib.tsm_debug_set_stamps(st.data_ptr())
with policy_within_training_step(algo):
    for _ in range(int(os.environ.get("STAMP_COLLECTS", "200"))):   # (the stamps kept are the last collect's: warm clocks)
        col.collect(n_step=E * T)
        col.reset_buffer(keep_statistics=True)
torch.cuda.synchronize()
lib.tsm_debug_set_stamps(None)
s = st.cpu().numpy()[:128].reshape(4, 32)
if form != "tile":
    names = ["index algebra", "obs fragments + obs store", "layer 1", "layer 2", "logits", "heads", "pair forces",
             "fold + integrate + publish", "reward terms + reward", "episode returns + stores + reset"]
    for t in range(1, 4):
        d = [(s[t][k + 1] - s[t][k]) / 100.0 for k in range(9)] + [(s[t + 1][0] - s[t][9]) / 100.0 if t < 3 else 0.0]
        print(f"step {t}: total {(s[t][9] - s[t][0]) / 100.0:.2f} us (wave 0 of workgroup 0; the other wave of its SIMD runs beside it)")
        print("   " + ", ".join(f"{n} {x:.2f}" for n, x in zip(names, d[:9])))
        print(f"   forward {sum(d[2:5]):.2f}, heads + env step {sum(d[5:9]):.2f}")
    sys.exit(0)
names = ["index algebra", "obs rows -> buffer + X(0)"] + \
        [f"tile {i} {l}" for i in range(4) for l in ("layer 1 (+ logits of the previous tile)", "X(next) + layer 2", "logits")] + \
        ["heads + pair forces", "-", "fold + integrate + publish", "reward terms + obs_next store", "reward", "stores", "reset / end"]
for t in range(1, 4):
    d = [(s[t][k + 1] - s[t][k]) / 100.0 for k in range(21)]
    print(f"step {t}: total {(s[t][21] - s[t][0]) / 100.0:.2f} us")
    print("   " + ", ".join(f"{n} {x:.2f}" for n, x in zip(names, d)))
    fw = sum(d[2:14])
    print(f"   forward of 4 tiles {fw:.2f}, heads + pair forces {d[14]:.2f}, rest of the env step {sum(d[15:21]):.2f}")
