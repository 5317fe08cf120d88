"""DeviceSimpleTagVectorEnv -- n_env PettingZoo-MPE `simple_tag` worlds (predators vs prey) stepped in one HIP kernel.

Stands where `DummyVectorEnv([lambda: EnhancedPettingZooEnv(simple_tag_v3.parallel_env())] * n)` stands in the
reference (/root/reference/tianshou/env/venvs.py:195-322, env/enhanced_pettingzoo_env.py:130-222) for the
two-team configurations (grouped policies, self-play / league trainers: BASELINE configs[4]).  Same vector-env contract
and the same two call styles as `DeviceSimpleSpreadVectorEnv` (env/mpe.py): `reset_device` / `step_device` on HBM
tensors for the Collector's device path, `reset` / `step` with numpy and the parallel-mode dict layout otherwise.
Agents: `adversary_0 .. adversary_{n_adv-1}`, then `agent_0 ..` (pettingzoo's order); `agent_groups` gives the team
split that `FlexibleMultiAgentPolicyManager(mode="grouped")` takes.  Observations are zero-padded to one width
(`obs_dim`), since the wrappers require identical spaces.  Dynamics follow the published MPE spec; parity with
pettingzoo itself is UNPINNED (csrc/mpe_tag.hip).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .._abi import call, ptr, stream_ptr, tsm_mpe_tag_cfg
from .spaces import Box, Discrete


class DeviceSimpleTagVectorEnv:
    is_async = False

    def __init__(self, n_env: int, num_good: int = 1, num_adversaries: int = 3, num_obstacles: int = 2,
                 max_cycles: int = 25, device: str | torch.device = "cuda", seed: int = 0, auto_reset: bool = True) -> None:
        self.env_num = int(n_env)
        self.n_adv, self.n_good, self.n_obst = int(num_adversaries), int(num_good), int(num_obstacles)
        self.n_agent = self.n_adv + self.n_good
        self.n_act = 5
        self.max_cycles = int(max_cycles)
        self.device = torch.device(device)
        self.auto_reset = auto_reset
        self.agents = [f"adversary_{i}" for i in range(self.n_adv)] + [f"agent_{i}" for i in range(self.n_good)]
        self.agent_idx = {a: i for i, a in enumerate(self.agents)}
        self.agent_groups = {"adversaries": self.agents[:self.n_adv], "good": self.agents[self.n_adv:]}
        self._tag_cfg = tsm_mpe_tag_cfg(self.env_num, self.n_adv, self.n_good, self.n_obst, self.max_cycles, 0,
                                        0.1, 0.25, 100.0, 1e-3, 0.075, 0.05, 0.2, 3.0, 4.0, 1.0, 1.3)
        self.obs_dim = call("tsm_mpe_tag_obs_dim", C.byref(self._tag_cfg))
        if self.obs_dim < 0:
            raise ValueError("simple_tag: unsupported sizes (at most 8 agents and 4 obstacles)")
        self.observation_space = Box(-np.inf, np.inf, (self.obs_dim,))
        self.action_space = [Discrete(self.n_act) for _ in range(self.env_num)]
        self._seed = int(seed)
        E, N, D, dev = self.env_num, self.n_agent, self.obs_dim, self.device
        f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)  # noqa: E731
        self.agent_pos, self.agent_vel, self.landmark_pos = f(E, N, 2), f(E, N, 2), f(E, max(1, self.n_obst), 2)
        self.steps = torch.zeros(E, dtype=torch.int32, device=dev)
        self.episode_ctr = torch.zeros(E, dtype=torch.int64, device=dev)
        self.rng_tick = torch.zeros(1, dtype=torch.int64, device=dev)
        self._obs_pp, self._pp = [f(E, N, D), f(E, N, D)], 0  # ping-pong: see env/mpe.py
        self.obs_next = f(E, N, D)
        self.rew = f(E, N)
        self.terminated = torch.zeros(E, N, dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros(E, N, dtype=torch.uint8, device=dev)
        self.done_env = torch.zeros(E, dtype=torch.uint8, device=dev)

    def __len__(self) -> int:
        return self.env_num

    @property
    def obs_cur(self) -> torch.Tensor:
        return self._obs_pp[self._pp]

    def seed(self, seed: int | None = None) -> list:
        if seed is not None:
            self._seed = int(seed)
            self.episode_ctr.zero_()
        return [self._seed + i for i in range(self.env_num)]

    def close(self) -> None:
        pass

    # ---- device path ------------------------------------------------------------------------
    def reset_device(self, env_ids: torch.Tensor | None = None) -> torch.Tensor:
        ids = None if env_ids is None else env_ids.to(self.device, torch.int64).contiguous()
        call("tsm_mpe_tag_reset", C.byref(self._tag_cfg), self._seed, ptr(self.episode_ctr), ptr(ids),
             0 if ids is None else ids.numel(), ptr(self.agent_pos), ptr(self.agent_vel), ptr(self.landmark_pos),
             ptr(self.steps), ptr(self.obs_cur), stream_ptr())
        return self.obs_cur

    def step_device(self, act: torch.Tensor, rng_tick_inc: int = 0):
        """One joint step; act i32 [n_env, n_agent].  Returns (obs_next, rew, terminated, truncated, done_env)."""
        if act.dtype != torch.int32:
            raise ValueError("step_device: act must be int32")
        self._pp ^= 1
        call("tsm_mpe_tag_step", C.byref(self._tag_cfg), self._seed, ptr(self.episode_ctr), ptr(act.contiguous()),
             ptr(self.agent_pos), ptr(self.agent_vel), ptr(self.landmark_pos), ptr(self.steps), ptr(self.obs_next),
             ptr(self.obs_cur), ptr(self.rew), ptr(self.terminated), ptr(self.truncated), ptr(self.done_env),
             int(self.auto_reset), ptr(self.rng_tick), int(rng_tick_inc), stream_ptr())
        return self.obs_next, self.rew, self.terminated, self.truncated, self.done_env

    # ---- reference-style (numpy) API --------------------------------------------------------
    def _obs_dicts(self, obs: np.ndarray, ids: np.ndarray) -> np.ndarray:
        out = np.empty(len(ids), dtype=object)
        for k, e in enumerate(ids):
            out[k] = {"observations": {a: obs[e, i] for i, a in enumerate(self.agents)}, "agent_ids": list(self.agents),
                      "masks": {a: [True] * self.n_act for a in self.agents}}
        return out

    def reset(self, env_id=None, **kwargs):
        ids = np.arange(self.env_num) if env_id is None else np.atleast_1d(np.asarray(env_id))
        obs = self.reset_device(None if env_id is None else torch.as_tensor(ids)).cpu().numpy()
        return self._obs_dicts(obs, ids), np.array([{"env_id": int(e)} for e in ids], dtype=object)

    def step(self, action, id=None):  # noqa: A002
        ids = np.arange(self.env_num) if id is None else np.atleast_1d(np.asarray(id))
        if len(ids) != self.env_num:
            raise ValueError("DeviceSimpleTagVectorEnv steps all envs together (synchronous vector env)")
        act = torch.as_tensor(np.asarray(action).reshape(self.env_num, self.n_agent)).to(self.device, torch.int32)
        keep, self.auto_reset = self.auto_reset, False  # the reference Collector resets finished envs itself
        try:
            obs_next, rew, term, trunc, _ = self.step_device(act)
        finally:
            self.auto_reset = keep
        info = np.array([{"env_id": int(e)} for e in ids], dtype=object)
        return (self._obs_dicts(obs_next.cpu().numpy(), ids), rew.cpu().numpy().astype(np.float64),
                term.cpu().numpy().astype(bool), trunc.cpu().numpy().astype(bool), info)
