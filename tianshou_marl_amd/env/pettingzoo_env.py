"""PettingZooEnv -- AEC (agent-environment-cycle) wrapper, host side.

Mirror of /root/reference/tianshou/env/pettingzoo_env.py:20-132.  Same construction checks (identical
per-agent spaces), same `reset() -> (obs_dict, info)` and `step(a) -> (obs_dict, rewards list indexed by
agent_idx, term, trunc, info)` formats, `agents` / `agent_idx` attributes the MARL dispatcher reads
(marl.py:197-203).  The wrapped env is duck-typed (pettingzoo is not a dependency here): it needs
`possible_agents`, `observation_space(agent)`, `action_space(agent)`, `reset()`, `last()`, `step()`,
`agent_selection`, `rewards`.

This wrapper is host plumbing for the reference's single-env configuration (BASELINE configs[0]); the
vectorised hot path uses the joint-step device layout instead (env/mpe.py), which is what
`EnhancedPettingZooEnv(mode="parallel")` emits per env.
"""
from __future__ import annotations

from typing import Any

from .spaces import is_discrete


class PettingZooEnv:
    def __init__(self, env: Any) -> None:
        self.env = env
        self.agents = list(env.possible_agents)
        self.agent_idx = {agent_id: i for i, agent_id in enumerate(self.agents)}
        self.rewards = [0] * len(self.agents)
        first = self.agents[0]
        self.observation_space: Any = env.observation_space(first)
        self.action_space: Any = env.action_space(first)
        if not all(env.observation_space(a) == self.observation_space for a in self.agents):
            raise AssertionError(
                "Observation spaces for all agents must be identical. Perhaps SuperSuit's pad_observations "
                "wrapper can help (usage: `supersuit.pad_observations_v0(env)`")
        if not all(env.action_space(a) == self.action_space for a in self.agents):
            raise AssertionError(
                "Action spaces for all agents must be identical. Perhaps SuperSuit's pad_action_space "
                "wrapper can help (usage: `supersuit.pad_action_space_v0(env)`")
        self.reset()

    # the three observation formats of pettingzoo_env.py:76-93 / :102-116
    def _observation_dict(self, observation: Any) -> dict:
        agent = self.env.agent_selection
        if isinstance(observation, dict) and "action_mask" in observation:
            return {"agent_id": agent, "obs": observation["observation"],
                    "mask": [m == 1 for m in observation["action_mask"]]}
        if is_discrete(self.action_space):
            return {"agent_id": agent, "obs": observation, "mask": [True] * self.env.action_space(agent).n}
        return {"agent_id": agent, "obs": observation}

    def reset(self, *args: Any, **kwargs: Any) -> tuple[dict, dict]:
        self.env.reset(*args, **kwargs)
        observation, _, _, _, info = self.env.last()
        return self._observation_dict(observation), info

    def step(self, action: Any) -> tuple[dict, list, bool, bool, dict]:
        self.env.step(action)
        observation, _, term, trunc, info = self.env.last()
        obs = self._observation_dict(observation)
        for agent_id, reward in self.env.rewards.items():
            self.rewards[self.agent_idx[agent_id]] = reward
        return obs, self.rewards, term, trunc, info

    def close(self) -> None:
        self.env.close()

    def seed(self, seed: Any = None) -> None:
        try:
            self.env.seed(seed)
        except (NotImplementedError, AttributeError):
            self.env.reset(seed=seed)

    def render(self) -> Any:
        return self.env.render()
