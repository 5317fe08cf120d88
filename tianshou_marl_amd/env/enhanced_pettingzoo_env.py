"""EnhancedPettingZooEnv -- AEC or Parallel PettingZoo wrapper, host side.

Mirror of /root/reference/tianshou/env/enhanced_pettingzoo_env.py:14-237: `mode` in {"aec", "parallel",
"auto"} (auto = parallel iff the env has `observation_spaces`), parallel `reset()` returns
`({"observations": {agent: obs}, "agent_ids": [...], "masks": {...}}, infos)` and `step(actions)` (dict or
array in `agents` order) returns `(obs_dict, reward list, term list, trunc list, infos)`, all lists indexed
by `agent_idx`.  The joint-step row this produces per env is exactly the row layout of the device buffer
(data/buffer.py); `rows_from_parallel_step` converts one such step to that layout.
"""
from __future__ import annotations

from typing import Any, Literal

import numpy as np

from .pettingzoo_env import PettingZooEnv
from .spaces import is_discrete


class EnhancedPettingZooEnv(PettingZooEnv):
    def __init__(self, env: Any, mode: Literal["aec", "parallel", "auto"] = "auto") -> None:
        if mode == "auto":
            mode = "parallel" if hasattr(env, "observation_spaces") else "aec"
        if mode not in ("aec", "parallel"):
            raise ValueError(f"unknown mode {mode!r}")
        self.mode = mode
        self.is_parallel = mode == "parallel"
        self.metadata = getattr(env, "metadata", {})
        if not self.is_parallel:
            super().__init__(env)
            self._num_agents = len(self.agents)
            return
        self.env = env
        self.agents = list(env.possible_agents)
        self.agent_idx = {agent: i for i, agent in enumerate(self.agents)}
        self.rewards = [0] * len(self.agents)
        self._num_agents = len(self.agents)
        first = self.agents[0]
        self.observation_space = env.observation_spaces[first]
        self.action_space = env.action_spaces[first]
        if not all(env.observation_spaces[a] == self.observation_space for a in self.agents):
            raise AssertionError("All agents must have identical observation spaces")
        if not all(env.action_spaces[a] == self.action_space for a in self.agents):
            raise AssertionError("All agents must have identical action spaces")
        self._seed = None

    @property
    def num_agents(self) -> int:
        return self._num_agents

    # ---- parallel mode ------------------------------------------------------------------------
    def _joint_observation(self, observations: dict) -> dict:
        out = {"observations": observations, "agent_ids": list(self.agents)}
        if is_discrete(self.action_space):
            n = self.action_space.n
            masks = {}
            for agent in self.agents:
                if agent not in observations:
                    masks[agent] = [False] * n  # terminated agent: nothing is legal
                elif isinstance(observations[agent], dict) and "action_mask" in observations[agent]:
                    masks[agent] = [m == 1 for m in observations[agent]["action_mask"]]
                else:
                    masks[agent] = [True] * n
            out["masks"] = masks
        return out

    def reset(self, *args: Any, **kwargs: Any) -> tuple[dict, dict]:
        if not self.is_parallel:
            return super().reset(*args, **kwargs)
        observations, infos = self.env.reset(*args, **kwargs)
        return self._joint_observation(observations), infos

    def step(self, action: Any):
        if not self.is_parallel:
            return super().step(action)
        if isinstance(action, (np.ndarray, list, tuple)):
            action = {agent: action[i] for i, agent in enumerate(self.agents)}
        observations, rewards, terminations, truncations, infos = self.env.step(action)
        n = len(self.agents)
        rew, term, trunc = [0.0] * n, [False] * n, [False] * n
        for agent, r in rewards.items():
            rew[self.agent_idx[agent]] = r
        for agent, t in terminations.items():
            term[self.agent_idx[agent]] = t
        for agent, t in truncations.items():
            trunc[self.agent_idx[agent]] = t
        return self._joint_observation(observations), rew, term, trunc, infos

    def seed(self, seed: Any = None) -> None:
        if self.is_parallel:
            self._seed = seed  # ParallelEnv seeds through reset(seed=...)
        else:
            super().seed(seed)


def rows_from_parallel_step(agents: list, obs_dict: dict, dtype=np.float32) -> np.ndarray:
    """One parallel-mode observation dict -> [N, D] in `agents` order (never dict order, quirk Q5)."""
    o = obs_dict["observations"]
    return np.stack([np.asarray(o[a]["observation"] if isinstance(o[a], dict) else o[a], dtype) for a in agents])
