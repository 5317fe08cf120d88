"""DeviceSimpleSpreadVectorEnv -- n_env PettingZoo-MPE `simple_spread` worlds stepped in one HIP kernel.

Stands where `DummyVectorEnv([lambda: EnhancedPettingZooEnv(simple_spread_v3.parallel_env())] * n)` stands
in the reference (/root/reference/tianshou/env/venvs.py:195-322, env/enhanced_pettingzoo_env.py:130-222):
same vector-env contract (`len(env)`, `reset(env_id)`, `step(action, id)`, `action_space`, `is_async`,
`seed`, `close`), same per-agent data (obs per agent, reward list, terminated/truncated lists), but the
joint state of every env lives in HBM and the AoS dict-of-agents is never materialised on the host.
Two call styles:
  * device path:  `reset_device()` / `step_device(act[n_env, N] i32)` -> HBM tensors (used by the Collector)
  * reference style: `reset(env_id)` / `step(action, id)` -> numpy, obs as the parallel-mode dict layout.
Dynamics follow the published MPE spec; parity with pettingzoo itself is UNPINNED (pettingzoo is neither
in the reference tree nor installed) -- see csrc/mpe.hip and DESIGN.md.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .._abi import call, ptr, stream_ptr, tsm_mpe_cfg
from .spaces import Box, Discrete


class DeviceSimpleSpreadVectorEnv:
    """simple_spread_v3(N, local_ratio=0.5, max_cycles=25, continuous_actions=False) x n_env on one GPU."""

    is_async = False

    def __init__(self, n_env: int, n_agent: int = 3, max_cycles: int = 25, local_ratio: float = 0.5,
                 device: str | torch.device = "cuda", seed: int = 0, auto_reset: bool = True) -> None:
        self.env_num = int(n_env)
        self.n_agent = int(n_agent)
        self.obs_dim = 6 * self.n_agent
        self.n_act = 5
        self.max_cycles = int(max_cycles)
        self.device = torch.device(device)
        self.auto_reset = auto_reset
        self.agents = [f"agent_{i}" for i in range(self.n_agent)]
        self.agent_idx = {a: i for i, a in enumerate(self.agents)}
        self.observation_space = Box(-np.inf, np.inf, (self.obs_dim,))
        self.action_space = [Discrete(self.n_act) for _ in range(self.env_num)]
        self._cfg = tsm_mpe_cfg(self.env_num, self.n_agent, self.max_cycles, 0, 0.1, 0.25, 100.0, 1e-3, 0.15, 0.05,
                                5.0, -1.0, float(local_ratio))
        self._seed = int(seed)
        E, N, D, dev = self.env_num, self.n_agent, self.obs_dim, self.device
        f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)  # noqa: E731
        self.agent_pos, self.agent_vel, self.landmark_pos = f(E, N, 2), f(E, N, 2), f(E, N, 2)
        self.steps = torch.zeros(E, dtype=torch.int32, device=dev)
        self.episode_ctr = torch.zeros(E, dtype=torch.int64, device=dev)  # u64 on the ABI
        self.rng_tick = torch.zeros(1, dtype=torch.int64, device=dev)
        # double-buffered outputs (static addresses: graph-capture friendly)
        # obs_cur ping-pongs between two buffers: the step kernel writes the NEXT policy input into the other
        # one, so the observation the policy just acted on stays intact until it has been added to the buffer
        self._obs_pp, self._pp = [f(E, N, D), f(E, N, D)], 0
        self.obs_next = f(E, N, D)
        self.rew = f(E, N)
        self.terminated = torch.zeros(E, N, dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros(E, N, dtype=torch.uint8, device=dev)
        self.done_env = torch.zeros(E, dtype=torch.uint8, device=dev)
        self._closed = False

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
        self._closed = True

    # ---- device path ------------------------------------------------------------------------
    def reset_device(self, env_ids: torch.Tensor | None = None) -> torch.Tensor:
        """Reset all (or the given) envs; returns obs_cur [n_env, N, obs_dim] (HBM, static storage)."""
        ids = None if env_ids is None else env_ids.to(self.device, torch.int64).contiguous()
        call("tsm_mpe_spread_reset", C.byref(self._cfg), self._seed, ptr(self.episode_ctr), ptr(ids),
             0 if ids is None else ids.numel(), ptr(self.agent_pos), ptr(self.agent_vel), ptr(self.landmark_pos),
             ptr(self.steps), ptr(self.obs_cur), stream_ptr())
        return self.obs_cur

    def step_device(self, act: torch.Tensor, rng_tick_inc: int = 0):
        """One joint step.  act i32 [n_env, N].  Returns (obs_next, rew, terminated, truncated, done_env) and
        flips `self.obs_cur` to the next policy input (reset observation where an episode finished and
        auto_reset is on); the previous `obs_cur` tensor keeps its contents."""
        if act.dtype != torch.int32:
            raise ValueError("step_device: act must be int32")
        self._pp ^= 1
        call("tsm_mpe_spread_step", C.byref(self._cfg), self._seed, ptr(self.episode_ctr), ptr(act.contiguous()),
             ptr(self.agent_pos), ptr(self.agent_vel), ptr(self.landmark_pos), ptr(self.steps), ptr(self.obs_next),
             ptr(self.obs_cur), ptr(self.rew), ptr(self.terminated), ptr(self.truncated), ptr(self.done_env),
             int(self.auto_reset), ptr(self.rng_tick), int(rng_tick_inc), stream_ptr())
        return self.obs_next, self.rew, self.terminated, self.truncated, self.done_env

    # ---- reference-style (numpy) API: BaseVectorEnv contract, venvs.py:195-322 ----------------
    def _obs_dicts(self, obs: np.ndarray, ids: np.ndarray) -> np.ndarray:
        out = np.empty(len(ids), dtype=object)
        for k, e in enumerate(ids):  # parallel-mode layout of enhanced_pettingzoo_env.py:202-220
            out[k] = {"observations": {a: obs[e, i] for i, a in enumerate(self.agents)},
                      "agent_ids": list(self.agents),
                      "masks": {a: [True] * self.n_act for a in self.agents}}
        return out

    def reset(self, env_id=None, **kwargs):
        ids = np.arange(self.env_num) if env_id is None else np.atleast_1d(np.asarray(env_id))
        obs = self.reset_device(None if env_id is None else torch.as_tensor(ids)).cpu().numpy()
        return self._obs_dicts(obs, ids), np.array([{"env_id": int(e)} for e in ids], dtype=object)

    def step(self, action, id=None):  # noqa: A002
        ids = np.arange(self.env_num) if id is None else np.atleast_1d(np.asarray(id))
        if len(ids) != self.env_num:
            raise ValueError("DeviceSimpleSpreadVectorEnv steps all envs together (synchronous vector env)")
        act = torch.as_tensor(np.asarray(action).reshape(self.env_num, self.n_agent)).to(self.device, torch.int32)
        keep = self.auto_reset
        self.auto_reset = False  # the reference Collector resets finished envs itself (collector.py:971)
        try:
            obs_next, rew, term, trunc, _ = self.step_device(act)
        finally:
            self.auto_reset = keep
        info = np.array([{"env_id": int(e)} for e in ids], dtype=object)
        return (self._obs_dicts(obs_next.cpu().numpy(), ids), rew.cpu().numpy().astype(np.float64),
                term.cpu().numpy().astype(bool), trunc.cpu().numpy().astype(bool), info)
