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


class LazyScalars(dict):
    """Named f32 scalars a kernel is writing into pinned host memory: a dict filled in when first read (the reference
    returns `.item()` floats, ctde.py:196-199; here the host need not wait for them to queue the next learner)."""

    def __init__(self, slot: dict, names: tuple) -> None:
        super().__init__()
        self._slot, self._names = slot, names

    def _force(self) -> None:
        slot = self._slot
        if slot is None:
            return
        self._slot = None
        slot["event"].synchronize()
        vals = slot["h"].numpy()
        dict.update(self, {k: float(vals[i]) for i, k in enumerate(self._names)})
        if slot.get("pending") is self:
            slot["pending"] = None

    def __getitem__(self, k):
        self._force()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        self._force()
        return dict.get(self, k, default)

    def __contains__(self, k):
        self._force()
        return dict.__contains__(self, k)

    def __iter__(self):
        self._force()
        return dict.__iter__(self)

    def __len__(self):
        self._force()
        return dict.__len__(self)

    def keys(self):
        self._force()
        return dict.keys(self)

    def values(self):
        self._force()
        return dict.values(self)

    def items(self):
        self._force()
        return dict.items(self)

    def __repr__(self):
        self._force()
        return dict.__repr__(self)


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
        # fused=True: learn() on rows readable in place runs on the one-launch kernels (_learn_store); async_stats=True:
        # its losses come back as a mapping that waits for the device only when it is read
        self.fused = bool(kwargs.pop("fused", True))
        self.async_stats = bool(kwargs.pop("async_stats", False))
        self.graph = bool(kwargs.pop("graph", True))  # the fused learn as one hipGraph replay per call
        self._ws: dict = {}
        self._cfg_pg = ops.make_ppo_cfg(adv_norm=False, ent_coef=0.0, vf_coef=0.0, loss_kind=1)

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

    # ---- fused learn: the stores read in place, 8 launches, no activation ever in HBM ------------------------------
    def _store_path(self, batch: Batch):
        """(store, agent) when this call can take the one-launch kernels: rows of a device buffer readable in place
        (`batch.chain_done.store`, training_coordinator.agent_batches_from_buffer), a centralized critic in-128-128-n_out on
        the concatenated joint row and a 128-wide actor."""
        if not self.fused or not self.enable_global_info or "chain_done" not in batch or "agent_index" not in batch:
            return None
        store = getattr(batch.chain_done, "store", None)
        if store is None:
            return None
        a, c = self.actor, self.critic
        ok = (c.dims[0] == store.N * store.D and a.dims[0] == store.D and len(a.dims) == 4 and len(c.dims) == 4
              and ops.critic_rows_grad_supported(c.dims[0], c.dims[1:-1], c.dims[-1], c.act)
              and ops.ppo_actor_rows_supported(a.dims[0], a.dims[1:-1], a.dims[-1], a.act)
              and not self.optim_actor.max_grad_norm and not self.optim_critic.max_grad_norm)
        return (store, int(batch.agent_index)) if ok else None

    def _learn_store(self, store, agent: int):
        """ctde.py:121-199 on the rows of `store` for agent column `agent`:
            v_last      critic on obs_next of the last slot                        tsm_critic_rows_forward
            critic      forward + TD loss + backward (values_next = the next row)  tsm_critic_rows_grad_td, tsm_critic_rows_dw1
            actor       forward + d(-mean log_probs) + backward                    tsm_ppo_actor_rows_update (loss_kind 1, adv = 1)
            scalars     actor_loss, critic_loss, mean(advantage)                   tsm_ctde_finalize
            two Adam steps (the actor's gradient scaled by mean(advantage) on the device: quirk Q7)   tsm_adam_step_segs
        The statistics land in pinned host memory behind the launches and are read when the returned mapping is."""
        T, E, N, D = store.T, store.E, store.N, store.D
        B, dev = T * E, self.device
        actor, critic = self.actor, self.critic
        H, n_out, A = critic.dims[1], critic.dims[-1], actor.dims[-1]
        K1 = N * D
        wkey = (store.key, T, agent, ops.kernel_options())  # (the options pick kernels and slab counts: captured per value)
        w = self._ws.get(wkey)
        if w is None:
            na = ops.ppo_actor_rows_grid(B)
            w = self._ws[wkey] = dict(
                ids=(torch.arange(B, dtype=torch.int64, device=dev) * N + agent).contiguous(), na=na,
                slabs_a=torch.empty(na, actor.flat.numel(), dtype=torch.float32, device=dev),
                part_a=torch.zeros(na * 4, dtype=torch.float64, device=dev),
                v_last=torch.empty(E, dtype=torch.float32, device=dev),
                # V(obs_next) of every row, env-major; filled only when an episode ended before the last slot (device flag)
                v_full=torch.empty(B, dtype=torch.float32, device=dev),
                em_rows=(torch.arange(B, dtype=torch.int64, device=dev) % T * E
                         + torch.div(torch.arange(B, dtype=torch.int64, device=dev), T, rounding_mode="floor")).contiguous(),
                mean_adv=torch.zeros(1, dtype=torch.float32, device=dev), ring=[], pos=0, calls=0, graphs={}, gslots={},
                step_dev=torch.zeros(1, dtype=torch.int64, device=dev))
        nW1 = H * K1

        def body(slot, step_dev):
            ops.critic_rows_forward(critic.flat.data, store.obs_next[T - 1].reshape(E, K1), H, n_out=n_out, out=w["v_last"])
            ops.critic_rows_forward(critic.flat.data, store.obs_next[:T].reshape(B, K1), H, n_out=n_out, rows=w["em_rows"],
                                    Mr=B, run_if=store.early_done, out=w["v_full"])  # (a no-op launch for aligned collects)
            w1s, rest, part_c = ops.critic_rows_grad_td(
                critic.flat.data, store.obs[:T].reshape(T, E, K1), T, E, store.rew[:T], store.term[:T], agent, N, w["v_last"],
                self.discount_factor, n_out, H, ws=self._ws, v_next_full=w["v_full"], use_full=store.early_done)
            ops.ppo_actor_rows_update(actor.flat.data, store.obs[:T].reshape(B * N, D), store.act[:T].reshape(-1), None, None,
                                      self._cfg_pg, A, actor.dims[1], perm=w["ids"], M=B, n_blocks=w["na"],
                                      slabs=w["slabs_a"], partial=w["part_a"], opt_step_dev=step_dev)
            ops.ctde_finalize(part_c, part_c.numel() // 4, w["part_a"], w["na"], B, slot["h"], w["mean_adv"])
            self.optim_critic.step_segs([(w1s, 0, nW1), (rest, nW1, critic.flat.numel() - nW1)], step_dev=step_dev)
            self.optim_actor.step_segs([(w["slabs_a"], 0, actor.flat.numel(), w["mean_adv"])], step_dev=step_dev)

        mk_slot = lambda: dict(h=torch.zeros(2, dtype=torch.float32).pin_memory(), event=torch.cuda.Event(), pending=None)  # noqa: E731
        w["calls"] += 1
        same_steps = self.optim_actor.step_count == self.optim_critic.step_count
        hyper = tuple((o.lr, tuple(o.betas), o.eps, o.weight_decay) for o in (self.optim_actor, self.optim_critic)) + \
            (self.discount_factor,)
        if w.get("hyper") != hyper:  # captured launches hold these as kernel arguments
            w["graphs"].clear()
            w["hyper"] = hyper
        if self.graph and w["calls"] >= 2 and same_steps:
            # ONE hipGraph replay per call (the first call of a shape runs eagerly: one-time kernel attributes).  Two graphs
            # per agent take turns, each with a pinned slot of its own for the statistics, so that a replay never overwrites
            # numbers the caller has not read yet; the optimizer step count lives in HBM (advanced by the actor kernel)
            k = w["calls"] & 1
            slot = w["gslots"].setdefault(k, mk_slot())
            if slot["pending"] is not None:
                slot["pending"]._force()
            if w.get("step_host") != self.optim_actor.step_count:
                w["step_dev"].fill_(self.optim_actor.step_count)
            if k not in w["graphs"]:
                g = torch.cuda.CUDAGraph()
                with ops.graph_capture(g):
                    body(slot, w["step_dev"])
                w["graphs"][k] = g
            w["graphs"][k].replay()
            self.optim_actor.step_count += 1
            self.optim_critic.step_count += 1
            w["step_host"] = self.optim_actor.step_count
        else:
            if len(w["ring"]) < 4:  # pinned slots the finalize kernel writes straight into (no copy on the stream)
                w["ring"].append(mk_slot())
            slot = w["ring"][w["pos"] % len(w["ring"])] if len(w["ring"]) == 4 else w["ring"][-1]
            w["pos"] += 1
            if slot["pending"] is not None:
                slot["pending"]._force()
            body(slot, None)
        slot["event"].record()
        out = LazyScalars(slot, ("actor_loss", "critic_loss"))
        slot["pending"] = out
        return out

    def learn(self, batch: Batch, **kwargs: Any) -> dict[str, float]:
        """One centralized-critic TD step + one policy-gradient step (ctde.py:121-199)."""
        fast = self._store_path(batch)
        if fast is not None:
            out = self._learn_store(*fast)
            return out if self.async_stats else dict(out)
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
