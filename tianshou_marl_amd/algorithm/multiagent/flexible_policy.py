"""FlexibleMultiAgentPolicyManager -- agent -> policy mapping with parameter sharing.

Mirror of /root/reference/tianshou/algorithm/multiagent/flexible_policy.py:12-253: modes "independent",
"shared", "grouped", "custom"; same validation errors; `policy_map` / `policy_mapping` (agent -> policy),
`policies` (unique policies keyed "shared" / group name / first agent), `get_shared_parameters()`,
`get_policy_groups()`.

Forward always routes through `policy_map` (agent -> policy).  The reference re-keys `self.policies` by group
name after construction (:94-98) and its inherited forward then compares agent ids with group names, so
"grouped" forward yields no `act` there (quirk Q6, recorded in tests/golden/marl_dispatch.npz); the documented
mapping semantics are what is implemented.  In "shared" mode all rows go through ONE fused forward launch
(the reference's `_forward_shared`, :202-231).
"""
from __future__ import annotations

from collections.abc import Callable
from typing import Any, Literal

import numpy as np
import torch

from ...data.batch import Batch
from .marl import MultiAgentPolicy


def _is_policy(p) -> bool:
    return callable(p) and not isinstance(p, list | dict)


class FlexibleMultiAgentPolicyManager(MultiAgentPolicy):
    def __init__(self, policies, env, mode: Literal["independent", "shared", "grouped", "custom"] = "independent",
                 agent_groups: dict[str, list[str]] | None = None,
                 policy_mapping_fn: Callable[[str], str] | None = None, **kwargs: Any) -> None:
        self.mode = mode
        self.agent_groups = agent_groups or {}
        self.policy_mapping_fn = policy_mapping_fn
        self._validate_config(policies)
        agents = list(env.agents)
        policy_map = self._build_policy_map(policies, agents)
        agent_idx = getattr(env, "agent_idx", None) or {a: i for i, a in enumerate(agents)}
        super().__init__(policies={a: policy_map[a] for a in agents}, agent_idx=agent_idx)
        self.env = env
        self.agents = agents
        self.policy_mapping = self.policy_map     # agent -> policy: what forward dispatches on
        self._original_policies = policies
        if mode == "shared":
            self.policies = {"shared": next(iter(self.policy_map.values()))}
        elif mode == "grouped":
            self.policies = {g: self.policy_map[members[0]] for g, members in self.agent_groups.items()}
        else:
            self.policies = {}
            for agent, policy in self.policy_map.items():
                if not any(policy is q for q in self.policies.values()):
                    self.policies[agent] = policy

    def _validate_config(self, policies) -> None:
        if self.mode == "independent":
            if _is_policy(policies):
                raise ValueError("Independent mode requires list or dict of policies, got single policy")
        elif self.mode == "grouped":
            if not self.agent_groups:
                raise ValueError("Grouped mode requires agent_groups to be specified")
            if _is_policy(policies):
                raise ValueError("Grouped mode requires dict of policies mapped to group names")
        elif self.mode == "custom":
            if not self.policy_mapping_fn:
                raise ValueError("Custom mode requires policy_mapping_fn to be specified")
        elif self.mode != "shared":
            raise ValueError(f"unknown mode {self.mode!r}")

    def _build_policy_map(self, policies, agents: list) -> dict:
        if self.mode == "shared":
            if isinstance(policies, list):
                shared = policies[0]
            elif isinstance(policies, dict):
                shared = next(iter(policies.values()))
            elif _is_policy(policies):
                shared = policies
            else:
                raise ValueError(f"Invalid policies type: {type(policies)}")
            return {agent: shared for agent in agents}
        if self.mode == "grouped":
            if not isinstance(policies, dict):
                raise ValueError("Grouped mode requires dict of policies")
            out = {}
            for group, members in self.agent_groups.items():
                if group not in policies:
                    raise ValueError(f"No policy found for group {group}")
                for agent in members:
                    if agent in agents:
                        out[agent] = policies[group]
            unassigned = set(agents) - set(out)
            if unassigned:
                raise ValueError(f"Agents {unassigned} are not assigned to any group")
            return out
        if self.mode == "custom":
            if not isinstance(policies, dict):
                raise ValueError("Custom mode requires dict of policies")
            out = {}
            for agent in agents:
                pid = self.policy_mapping_fn(agent)
                if pid not in policies:
                    raise ValueError(f"Policy {pid} not found for agent {agent}")
                out[agent] = policies[pid]
            return out
        if isinstance(policies, list):
            if len(policies) != len(agents):
                raise ValueError(f"Number of policies ({len(policies)}) must match number of agents ({len(agents)})")
            return dict(zip(agents, policies, strict=True))
        if isinstance(policies, dict):
            missing = set(agents) - set(policies)
            if missing:
                raise ValueError(f"Missing policies for agents: {missing}")
            return policies
        raise ValueError("Independent mode requires list or dict of policies")

    def forward(self, batch: Batch, state=None, **kwargs: Any) -> Batch:
        obs = batch.obs
        if self.mode == "shared" and isinstance(obs, Batch) and "agent_id" in obs:
            return self._forward_shared(batch, state, **kwargs)
        return super().forward(batch, state, **kwargs)

    def _forward_shared(self, batch: Batch, state=None, **kwargs: Any) -> Batch:
        shared = next(iter(self.policy_map.values()))
        obs = batch.obs
        rows = obs.obs if "obs" in obs else obs
        if not isinstance(rows, torch.Tensor | Batch):
            rows = np.asarray(rows)
        out = shared(Batch(obs=rows), state, **kwargs)  # ONE launch for all agents' rows
        holder = Batch(act=out.act)
        holder["out"] = {a: out for a in self.agents}
        holder["state"] = {a: (out.state if "state" in out and out.state is not None else Batch()) for a in self.agents}
        return holder

    def get_shared_parameters(self) -> bool:
        return len({id(p) for p in self.policy_map.values()}) < len(self.policy_map)

    def get_policy_groups(self) -> dict[str, list[str]]:
        by_policy: dict[int, list[str]] = {}
        for agent, policy in self.policy_map.items():
            by_policy.setdefault(id(policy), []).append(agent)
        if len(by_policy) == 1:
            return {"shared": next(iter(by_policy.values()))}
        return {f"group_{i}": members for i, members in enumerate(by_policy.values())}
