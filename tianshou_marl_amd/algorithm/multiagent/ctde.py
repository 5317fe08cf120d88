"""CTDE (centralized training, decentralized execution) on the HIP path.

Mirror of /root/reference/tianshou/algorithm/multiagent/ctde.py for the components of SURVEY section 8a row a16:
  `CTDEPolicy`               :22-227   forward = decentralized actor on local obs; learn = centralized critic TD
  `GlobalStateConstructor`   :230-343  "concatenate" / "mean" (+ "custom"); attention / graph aggregation are
                                       learned modules outside the hot path (out of scope)
  `DecentralizedActor`       :346-379  obs -> H -> H -> action logits, ReLU
  `CentralizedCritic`        :382-414  global_obs -> H -> H -> n_agents values, ReLU
QMIX / MADDPG (:417-954) are off-policy value-decomposition / continuous-control algorithms outside the
north-star path and are not built.

Networks are `FlatMLP`s (one flat HBM parameter vector each, csrc/dense.hip f32-MFMA GEMMs for forward, dgrad
and wgrad); the TD-target / MSE / policy-gradient head between them is `tsm_ctde_td_head` (csrc/ctde.hip) and the
two optimizer steps are `tsm_adam_step`.  `learn` reproduces the reference arithmetic exactly, including the
(B,) x (B,1) broadcast of ctde.py:185 (quirk Q7: actor_loss = -mean(log_probs) * mean(advantage)); parity with the
reference's losses, gradients and post-step weights is pinned by tests/golden/ctde.npz.
"""
from __future__ import annotations

from collections.abc import Callable
from typing import Any, Literal

import numpy as np
import torch
from torch import nn

from ... import ops
from ...data.batch import Batch
from ...utils.net import FlatAdam, FlatMLP


class DecentralizedActor(FlatMLP):
    """ctde.py:346-379: local observation -> action logits; `forward(obs, state) -> (logits, state)`."""

    def __init__(self, obs_dim: int, action_dim: int, hidden_dim: int = 128, device: str | torch.device = "cuda",
                 seed: int | None = None) -> None:
        super().__init__([obs_dim, hidden_dim, hidden_dim, action_dim], act="relu", device=device, seed=seed)

    def forward(self, obs: torch.Tensor, state=None, save: bool = False):  # type: ignore[override]
        return super().forward(obs, save=save), state


class CentralizedCritic(FlatMLP):
    """ctde.py:382-414: global state -> one value per agent."""

    def __init__(self, global_obs_dim: int, n_agents: int, hidden_dim: int = 128, device: str | torch.device = "cuda",
                 seed: int | None = None) -> None:
        super().__init__([global_obs_dim, hidden_dim, hidden_dim, n_agents], act="relu", device=device, seed=seed)


class GlobalStateConstructor(nn.Module):
    def __init__(self, mode: Literal["concatenate", "mean", "attention", "graph", "custom"] = "concatenate",
                 obs_dim: int | None = None, n_agents: int | None = None, hidden_dim: int = 64,
                 adjacency_matrix: torch.Tensor | None = None, custom_fn: Callable | None = None) -> None:
        super().__init__()
        if mode in ("attention", "graph"):
            raise NotImplementedError(
                f"GlobalStateConstructor(mode={mode!r}) is a learned aggregation outside the north-star hot path "
                "(SURVEY.md section 2); use 'concatenate', 'mean' or 'custom'")
        self.mode, self.obs_dim, self.n_agents = mode, obs_dim, n_agents
        self.adjacency_matrix, self.custom_fn = adjacency_matrix, custom_fn

    def build(self, observations: dict[str, torch.Tensor]) -> torch.Tensor:
        """observations: agent_id -> [B, D] (dict order = env.agents order; never a set, quirk Q5)."""
        if self.mode == "custom" and self.custom_fn:
            return self.custom_fn(observations)
        obs = [torch.as_tensor(np.asarray(o) if not isinstance(o, torch.Tensor) else o) for o in observations.values()]
        obs = [o.to("cuda", torch.float32).contiguous() if not o.is_cuda else o.to(torch.float32).contiguous() for o in obs]
        return ops.global_state(obs, "mean" if self.mode == "mean" else "concatenate")  # unknown modes concatenate (:340-343)

    @staticmethod
    def from_joint_rows(obs: torch.Tensor, mode: str = "concatenate") -> torch.Tensor:
        """Device layout shortcut: obs [R, N, D] (one joint step per row).  The concatenation is a free view."""
        if mode == "concatenate":
            return obs.reshape(obs.shape[0], -1)
        return ops.global_state([obs[:, a].contiguous() for a in range(obs.shape[1])], "mean")


class CTDEPolicy(nn.Module):
    def __init__(self, actor: FlatMLP, critic: FlatMLP, optim_actor: FlatAdam | None = None,
                 optim_critic: FlatAdam | None = None, observation_space: Any = None, action_space: Any = None,
                 enable_global_info: bool = True, discount_factor: float = 0.99, **kwargs: Any) -> None:
        super().__init__()
        self.tau = kwargs.pop("tau", 0.005)
        if not isinstance(actor, FlatMLP) or not isinstance(critic, FlatMLP):
            raise TypeError("CTDEPolicy needs FlatMLP networks (DecentralizedActor / CentralizedCritic): the update "
                            "runs in HIP, there is no autograd fallback")
        self.actor, self.critic = actor, critic
        self.optim_actor = optim_actor or FlatAdam(actor)
        self.optim_critic = optim_critic or FlatAdam(critic)
        self.observation_space, self.action_space = observation_space, action_space
        self.enable_global_info = enable_global_info
        self.discount_factor = discount_factor
        self.is_within_training_step = False
        self.deterministic_eval = bool(kwargs.pop("deterministic_eval", False))
        self.seed = int(kwargs.pop("seed", 0))
        self._sample_ctr = 0

    @property
    def device(self) -> torch.device:
        return self.actor.flat.device

    def _t(self, x, dtype) -> torch.Tensor:
        t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
        return t.to(self.device, dtype).contiguous()

    def forward(self, batch: Batch, state: Any = None, **kwargs: Any) -> Batch:
        """Decentralized execution (ctde.py:85-119): `act` holds the actor's raw output (logits), as upstream."""
        logits, state = self.actor(self._t(batch.obs, torch.float32), state)
        return Batch(act=logits, state=state)

    # ---- device rollout entry (Collector device path) ----------------------------------------------
    def act_device(self, obs: torch.Tensor, out: dict | None = None, offset_dev: torch.Tensor | None = None,
                   row_offset: int = 0) -> dict:
        """obs [..., D] in HBM -> dict(act i32, logp, value) per row: actor forward (dense.hip) + Categorical sample /
        mode (categorical.hip).  The reference leaves turning the actor's logits into an action to the caller
        (CTDEPolicy.forward returns the raw logits as `act`, ctde.py:119); for a Discrete action space that is a
        Categorical(logits) draw.  `value` is 0: the centralized critic needs the global state, which only the
        update sees."""
        rows = obs.reshape(-1, self.actor.dims[0])
        logits = FlatMLP.forward(self.actor, rows, save=False)
        greedy = bool(getattr(self, "deterministic_eval", False) and not self.is_within_training_step)
        res = (out["act"], out["logp"]) if out is not None else None
        act, logp = ops.categorical_sample(logits, self.seed, offset=self._sample_ctr + row_offset, deterministic=greedy,
                                           offset_dev=offset_dev, out=res)
        if offset_dev is None:
            self._sample_ctr += rows.shape[0]
        if out is not None:
            out["value"].zero_()
            return out
        return dict(act=act, logp=logp, value=torch.zeros_like(logp), logits=logits)

    def learn(self, batch: Batch, **kwargs: Any) -> dict[str, float]:
        """One centralized-critic TD step + one policy-gradient step (ctde.py:121-199)."""
        obs = self._t(batch.obs, torch.float32)
        act = self._t(batch.act, torch.int64).reshape(-1)
        rew = self._t(batch.rew, torch.float32).reshape(-1)
        obs_next = self._t(batch.obs_next, torch.float32)
        terminated = self._t(batch.terminated, torch.uint8).reshape(-1)
        B = obs.shape[0]
        if self.enable_global_info and "global_obs" in batch:
            critic_in, critic_in_next = self._t(batch.global_obs, torch.float32), self._t(batch.global_obs_next, torch.float32)
        else:
            critic_in, critic_in_next = obs, obs_next
        q = self.critic(critic_in, save=True).reshape(B, -1)
        chain = batch.chain_done if "chain_done" in batch else None
        T = getattr(chain, "chain_T", 0)
        if isinstance(chain, torch.Tensor) and chain.is_cuda and T >= 1 and chain.numel() == B and B % T == 0:
            # The TD target uses the ONLINE critic on obs_next (no target network in `learn`, ctde.py:165-172), and for
            # chained rows obs_next of (env, t) is obs of (env, t + 1): its values are rows of `q`, except at the last
            # step of each env's block -- a pass over those E rows -- and where an episode ended early: then (device flag)
            # the full pass runs.  Bit-identical to evaluating the critic on obs_next; saves a third of the critic work.
            E = B // T
            n_out = q.shape[1]
            flag = ops.any_nonzero_u8(chain.view(E, T)[:, :T - 1].contiguous().reshape(-1)) if T > 1 else \
                torch.zeros(1, dtype=torch.int32, device=q.device)
            q_full, _ = ops.mlp_forward_cond(self.critic.desc, self.critic.flat.data, critic_in_next, flag)
            q_last = FlatMLP.forward(self.critic, critic_in_next.view(E, T, -1)[:, T - 1].contiguous(), save=False)
            q_next = ops.value_next_select_env_major(q.contiguous(), q_last.reshape(E, n_out), q_full.reshape(B, n_out), flag,
                                                     E, T, n_out)
        else:
            q_next = self.critic(critic_in_next, save=False).reshape(B, -1)  # target side: no gradient (detach, :172)
        logits = FlatMLP.forward(self.actor, obs, save=True)
        dq, dlogits, scalars = ops.ctde_td_head(q, q_next, rew, terminated, self.discount_factor, logits, act)
        self.optim_critic.zero_grad()
        self.optim_critic.step(self.critic.backward(dq))
        self.optim_actor.zero_grad()
        self.optim_actor.step(self.actor.backward(dlogits))
        s = scalars.cpu().numpy()  # the two `.item()` of ctde.py:196-199 as one copy
        return {"actor_loss": float(s[0]), "critic_loss": float(s[1])}

    def soft_update_targets(self) -> None:
        """ctde.py:201-209: Polyak averaging into `actor_target` / `critic_target` when present."""
        for name in ("actor", "critic"):
            target = getattr(self, name + "_target", None)
            if target is not None:
                target.flat.data.mul_(1 - self.tau).add_(getattr(self, name).flat.data, alpha=self.tau)

    def _build_global_state(self, batch: Batch) -> torch.Tensor:
        return batch.global_obs if "global_obs" in batch else batch.obs

    def state_dict(self, *args, **kwargs):
        """The reference's CTDEPolicy is a plain nn.Module over `actor` and `critic` (ctde.py:346-414: fc1..fc3), so its
        state_dict is `actor.fc{i}.weight|bias`, `critic.fc{i}.weight|bias` and nothing else; the two optimizers are
        saved by their owner (`optim_actor.state_dict()`: torch-Adam layout, utils/net.FlatAdam).  Key names and shapes
        are pinned by tests/golden/checkpoint.npz."""
        from collections import OrderedDict

        sd = OrderedDict()
        for name, net in (("actor", self.actor), ("critic", self.critic)):
            for k, v in net.to_reference_state_dict().items():
                sd[f"{name}.{k}"] = v
        return sd

    def load_state_dict(self, sd, *args, **kwargs):
        for name, net in (("actor", self.actor), ("critic", self.critic)):
            net.load_reference_state_dict({k[len(name) + 1:]: v for k, v in sd.items() if k.startswith(name + ".")})
