#!/usr/bin/env python3
"""Phase time stamps of the persistent tag rollout (csrc/rollout_tag.hip) at the bench shard (512 worlds, 3 v 1): workgroup 0,
vector steps 1-3 (100 MHz wall clock).

    python tools/stamp_rollout_tag.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi  # noqa: E402
from tianshou_marl_amd.algorithm.multiagent import FlexibleMultiAgentPolicyManager  # noqa: E402
from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step  # noqa: E402
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer  # noqa: E402
from tianshou_marl_amd.data.collector import Collector  # noqa: E402
from tianshou_marl_amd.env.mpe_tag import DeviceSimpleTagVectorEnv  # noqa: E402
from tianshou_marl_amd.utils.net import DiscreteActorCritic  # noqa: E402

dev, E, T = "cuda", 512, 25
env = DeviceSimpleTagVectorEnv(E, device=dev, seed=1, max_cycles=T)
mk = lambda s: PPO(net=DiscreteActorCritic(env.obs_dim, 5, 64, device=dev, seed=s), seed=s)  # noqa: E731
mgr = FlexibleMultiAgentPolicyManager({"adversaries": mk(1), "good": mk(2)}, env, mode="grouped", agent_groups=env.agent_groups)
buf = DeviceVectorReplayBuffer(E * T, E, env.n_agent, env.obs_dim, device=dev)
col = Collector(mgr, env, buf)
col.reset()
st = torch.zeros(1024, dtype=torch.int64, device=dev)
lib = _abi.load()
lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]
lib.tsm_debug_set_stamps(st.data_ptr())
with policy_within_training_step(mgr):
    for _ in range(3):
        col.collect(n_step=E * T)
        col.reset_buffer(keep_statistics=True)
torch.cuda.synchronize()
lib.tsm_debug_set_stamps(None)
full = st.cpu().numpy()
print(f"workgroup 0: prologue (weights of both teams, env state, first observation rows) {(full[41] - full[40]) / 100.0:.2f} us, "
      f"{T} steps {(full[42] - full[41]) / 100.0:.2f} us")
s = full[:32].reshape(4, 8)
names = ["forward x 2 teams", "heads", "env step (move, publish, rewards, obs_next)", "payload scatter", "done / reset"]
for t in range(1, 4):
    d = [(s[t][k + 1] - s[t][k]) / 100.0 for k in range(5)]
    print(f"step {t}: total {(s[t][5] - s[t][0]) / 100.0:.2f} us: " + ", ".join(f"{n} {x:.2f}" for n, x in zip(names, d)))
