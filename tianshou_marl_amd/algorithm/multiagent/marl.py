"""Multi-agent policy container and per-agent dispatch of algorithms.

Mirror of /root/reference/tianshou/algorithm/multiagent/marl.py:
  `MultiAgentPolicy(policies)`            :77-185   forward() routes rows to the owning agent's policy
  `MARLDispatcher(algorithms, env)`       :191-268  per-agent `_preprocess_batch` / `_update_with_batch`
  `MultiAgentOnPolicyAlgorithm(...)`      :314-353
  `MapTrainingStats`                      :29-59    (data/stats.py)

Two row formats reach `forward`:
  * AEC rows (reference layout): `batch.obs = Batch(agent_id=[B], obs=[B, D][, mask])`, one agent per row.
    The row partition `np.nonzero(batch.obs.agent_id == agent)` (:148) is computed on the GPU by a stable
    counting sort (`tsm_agent_index`, csrc/dispatch.hip) and rows travel with `tsm_gather_rows` /
    `tsm_scatter_rows`; results are bit-identical to the reference's index sets (tests/golden/marl_dispatch.npz).
  * joint-step rows (device layout): obs `[R, N, D]`, agent a owns column a -- the
    partition is a stride, no index kernel is needed.
"""
from __future__ import annotations

from collections.abc import Callable
from typing import Any

import numpy as np
import torch
from torch import nn

from ... import ops
from ...data.batch import Batch
from ...data.stats import MapTrainingStats


def _unique(objs) -> list:
    seen, out = set(), []
    for o in objs:
        if id(o) not in seen:
            seen.add(id(o))
            out.append(o)
    return out


class MultiAgentPolicy(nn.Module):
    def __init__(self, policies: dict, agent_idx: dict | None = None) -> None:
        super().__init__()
        if len(policies) == 0:
            raise ValueError("MultiAgentPolicy needs at least one policy")
        self.policies = dict(policies)
        self.agent_idx = dict(agent_idx) if agent_idx is not None else {a: i for i, a in enumerate(self.policies)}
        p0 = next(iter(self.policies.values()))
        self.action_space = getattr(p0, "action_space", None)
        self.observation_space = getattr(p0, "observation_space", None)
        self.policy_map = self.policies  # agent -> policy; subclasses may re-key `policies` (flexible_policy.py:94-105)
        self._submodules = nn.ModuleList([p for p in _unique(self.policies.values()) if isinstance(p, nn.Module)])

    @property
    def _agent_policies(self) -> dict:
        return self.policy_map

    # ---- flags the collector / trainer toggle are forwarded to every sub-policy ----------------
    @property
    def is_within_training_step(self) -> bool:
        return all(getattr(p, "is_within_training_step", False) for p in self._agent_policies.values())

    @is_within_training_step.setter
    def is_within_training_step(self, v: bool) -> None:
        for p in _unique(self._agent_policies.values()):
            p.is_within_training_step = v

    @property
    def shared_policy(self):
        """The single policy object when every agent maps to the same one (parameter sharing), else None."""
        u = _unique(self._agent_policies.values())
        return u[0] if len(u) == 1 else None

    @property
    def device(self) -> torch.device:
        return next(iter(self._agent_policies.values())).device

    def _codes(self, agent_id) -> np.ndarray:
        ids = np.asarray(agent_id)
        codes = np.full(ids.shape, len(self.agent_idx), np.int32)  # unknown ids match no policy (as in :148)
        for a, k in self.agent_idx.items():
            codes[ids == a] = k
        return codes

    def _partition(self, agent_id):
        """(index i64[B] grouped by agent code, host offsets) via the HIP counting sort."""
        codes = torch.as_tensor(self._codes(agent_id)).to(self.device)
        index, offsets = ops.agent_index(codes, len(self.agent_idx) + 1)
        return index, offsets.cpu().numpy()

    def map_action(self, act):
        return act

    def add_exploration_noise(self, act, batch):
        """marl.py:91-106."""
        if not isinstance(batch.obs, Batch):
            raise TypeError(f"here only observations of type Batch are permitted, but got {type(batch.obs)}")
        ids = np.asarray(batch.obs.agent_id)
        for agent_id, policy in self._agent_policies.items():
            rows = np.nonzero(ids == agent_id)[0]
            if len(rows) == 0:
                continue
            act[rows] = policy.add_exploration_noise(act[rows], batch[rows])
        return act

    # ---- device rollout entry (Collector device path) --------------------------------------------
    def act_device(self, obs: torch.Tensor, out: dict | None = None, offset_dev: torch.Tensor | None = None) -> dict:
        """Joint-step rows obs[E, N, D] -> dict(act i32, logp, value) each flat [E*N] (agent a = column a)."""
        shared = self.shared_policy
        if shared is not None:
            return shared.act_device(obs, out=out, offset_dev=offset_dev)
        E, N = obs.shape[0], obs.shape[1]
        if out is None:
            dev = obs.device
            out = dict(act=torch.empty(E * N, dtype=torch.int32, device=dev), logp=torch.empty(E * N, device=dev),
                       value=torch.empty(E * N, device=dev), logits=None)
        # Each agent column gets its own sampling-counter range [a * E, (a + 1) * E) (a policy shared by a team must not
        # reuse draws).  Consecutive columns served by the same policy object go through ONE call on agent-major rows
        # (row j * E + e -> counter (a0 + j) * E + e: the same counters, hence the same draws, as column-by-column calls).
        for a0, k, policy in self._policy_runs():
            if k == 1:
                res = policy.act_device(obs[:, a0].contiguous(), out=None, offset_dev=offset_dev, row_offset=a0 * E)
                for f in ("act", "logp", "value"):
                    out[f].view(E, N)[:, a0].copy_(res[f].view(E))
            else:
                rows = obs[:, a0:a0 + k].transpose(0, 1).contiguous().view(k * E, -1)
                res = policy.act_device(rows, out=None, offset_dev=offset_dev, row_offset=a0 * E)
                for f in ("act", "logp", "value"):
                    out[f].view(E, N)[:, a0:a0 + k].copy_(res[f].view(k, E).transpose(0, 1))
        return out

    def _policy_runs(self) -> list:
        """[(first agent column, run length, policy)] for maximal runs of consecutive columns sharing a policy object."""
        # recomputed per call: trainers swap entries of the policy map (self-play opponents, league snapshots)
        by_col = sorted(((self.agent_idx[aid], pol) for aid, pol in self._agent_policies.items()), key=lambda t: t[0])
        runs: list = []
        for a, pol in by_col:
            if runs and runs[-1][2] is pol and runs[-1][0] + runs[-1][1] == a:
                runs[-1] = (runs[-1][0], runs[-1][1] + 1, pol)
            else:
                runs.append((a, 1, pol))
        return runs

    # ---- reference forward -------------------------------------------------------------------------
    def forward(self, batch: Batch, state: dict | Batch | None = None, **kwargs: Any) -> Batch:
        obs = batch.obs
        if isinstance(obs, Batch) and "agent_id" in obs:
            return self._forward_aec(batch, state, **kwargs)
        return self._forward_joint(batch, state, **kwargs)

    def _forward_joint(self, batch: Batch, state, **kwargs: Any) -> Batch:
        from ...data.buffer import _obs_array

        obs = batch.obs
        obs_t = obs if isinstance(obs, torch.Tensor) else torch.as_tensor(_obs_array(obs))
        obs_t = obs_t.to(self.device, torch.float32)
        if obs_t.dim() != 3 or obs_t.shape[1] != len(self.agent_idx):
            raise ValueError(f"joint-step observations must be [rows, {len(self.agent_idx)}, obs_dim], got {tuple(obs_t.shape)}")
        res = self.act_device(obs_t)
        R, N = obs_t.shape[0], obs_t.shape[1]
        holder = Batch(act=res["act"].view(R, N).to(torch.int64),
                       policy=Batch(logp=res["logp"].view(R, N), v_s=res["value"].view(R, N)))
        holder["state"] = {a: Batch() for a in self._agent_policies}
        return holder

    def _forward_aec(self, batch: Batch, state, **kwargs: Any) -> Batch:
        obs = batch.obs
        index, offs = self._partition(obs.agent_id)
        dev = self.device
        has_mask = "mask" in obs
        obs_rows = obs.obs if "obs" in obs else None
        obs_dev = None if obs_rows is None else torch.as_tensor(np.asarray(obs_rows) if not isinstance(
            obs_rows, torch.Tensor) else obs_rows).to(dev)
        B = len(np.asarray(obs.agent_id))
        act_holder = None
        out_dict, state_dict = {}, {}
        for agent_id, policy in self._agent_policies.items():
            k = self.agent_idx[agent_id]
            lo, hi = int(offs[k]), int(offs[k + 1])
            if hi == lo:  # no data for this agent (:149-152)
                out_dict[agent_id], state_dict[agent_id] = Batch(), Batch()
                continue
            rows = index[lo:hi]
            rows_h = rows.cpu().numpy()
            sub_obs = ops.gather_rows(obs_dev, rows)
            tmp = Batch(obs=sub_obs)
            if has_mask:  # the sub-policy sees the whole observation Batch when a mask is present (:157-161)
                tmp = Batch(obs=Batch(obs=sub_obs, mask=np.asarray(obs.mask)[rows_h],
                                      agent_id=np.asarray(obs.agent_id)[rows_h]))
            if "rew" in batch and isinstance(batch.rew, np.ndarray) and batch.rew.ndim == 2:
                tmp.rew = batch.rew[rows_h, k]  # :154-156
            if "info" in batch and isinstance(batch.info, np.ndarray | Batch) and len(batch.info) == B:
                tmp.info = batch.info[rows_h]
            out = policy(tmp, None if state is None else state[agent_id], **kwargs)
            act = out.act
            act_t = act if isinstance(act, torch.Tensor) else torch.as_tensor(np.asarray(act))
            act_t = act_t.to(dev)
            if act_holder is None:
                act_holder = torch.zeros((B, *act_t.shape[1:]), dtype=act_t.dtype, device=dev)
            ops.scatter_rows(act_t, rows, act_holder)  # holder.act[agent_index] = act (:180)
            each_state = out.state if ("state" in out and out.state is not None) else Batch()
            out_dict[agent_id], state_dict[agent_id] = out, each_state
        holder = Batch()
        if act_holder is not None:
            holder["act"] = act_holder
        holder["out"] = out_dict
        holder["state"] = state_dict
        return holder


class MARLDispatcher:
    """marl.py:191-268 on the device buffer: per-agent preprocessing and per-agent updates."""

    def __init__(self, algorithms: list, env) -> None:
        agent_ids = list(env.agents)
        assert len(algorithms) == len(agent_ids), "One policy must be assigned for each agent."
        self.algorithms: dict = dict(zip(agent_ids, algorithms, strict=True))
        self.agent_idx: dict = dict(env.agent_idx)

    def create_policy(self) -> MultiAgentPolicy:
        return MultiAgentPolicy({a: alg.policy for a, alg in self.algorithms.items()}, self.agent_idx)

    def dispatch_process_fn(self, buffer) -> dict:
        """agent_id -> preprocessed batch of that agent's algorithm (critic passes, GAE, logp_old).

        The reference slices the rows of each agent out of the flat buffer and temporarily swaps `buffer.rew`
        for that agent's reward column (:227-248); with joint-step lanes every agent's lane already carries
        its own reward/value stream, so one pass per distinct algorithm object covers all of its agents."""
        if getattr(buffer, "aec", False):
            return self._dispatch_process_aec(buffer)
        done: dict[int, dict] = {}
        results = {}
        for agent, algorithm in self.algorithms.items():
            if id(algorithm) not in done:
                done[id(algorithm)] = algorithm._preprocess_batch(buffer)
            results[agent] = done[id(algorithm)]
        return results

    # ---- AEC rows (DeviceAECReplayBuffer): the reference's own flat semantics, quirks included -------------------
    @staticmethod
    def aec_partition(buffer):
        """(indices i64 [n] = sample_indices(0), positions grouped by agent, host offsets [N + 1]):
        `np.nonzero(batch.obs.agent_id == agent)` of marl.py:233 for every agent, by one stable counting sort on device."""
        idx_all = buffer.index.sample_indices_all()
        codes = buffer._gather(buffer.agent_store.unsqueeze(-1), idx_all).view(-1)
        pos, offs = ops.agent_index(codes, len(buffer.agents))
        return idx_all, pos, offs.cpu().numpy()

    @staticmethod
    def aec_returns(buffer, tind: torch.Tensor, k: int, v_s: torch.Tensor, v_next: torch.Tensor, gamma: float,
                    gae_lambda: float, algorithm=None):
        """`compute_episodic_return` (algorithm_base.py:651-717) on ONE agent's rows `tind` (flat indices in sample(0)
        order) with that agent's reward column (marl.py:231,240): the filtered rows form one flat series whose end flags
        are terminated | truncated | isin(index, unfinished_index()) -- i.e. quirk Q1 is kept:
        a row that is the last of ITS agent in a sub-buffer but not the sub-buffer's last row carries the GAE of the next
        sub-buffer.  value_mask = ~terminated.  -> (returns, adv) f32 [n]."""
        n = tind.numel()
        g = lambda store: buffer._gather(store, tind)  # noqa: E731
        rew = g(buffer.rew_store)[:, k].contiguous().view(n, 1)
        term = g(buffer.term_store).view(n, 1)
        trunc = g(buffer.trunc_store).view(n, 1)
        forced = torch.zeros(buffer.maxsize, dtype=torch.uint8, device=tind.device)
        forced[buffer.index.unfinished_index()] = 1
        end = (trunc | forced[tind].view(n, 1)).contiguous()
        if algorithm is not None:  # through the algorithm: return_scaling statistics included (a2c.py:132-146)
            ret, adv = algorithm._gae(v_s.view(n, 1), v_next.view(n, 1), rew, term, end, 1)
        else:
            ret, adv = ops.gae_lanes(v_s.view(n, 1), v_next.view(n, 1), rew, term, end, gamma, gae_lambda)
        return ret.view(n), adv.view(n)

    def _dispatch_process_aec(self, buffer) -> dict:
        """marl.py:208-249 on AEC rows: per agent, slice its rows (and their flat indices) out of the batch, take its
        reward column, and run the sub-algorithm's `_preprocess_batch` arithmetic on them (critic on obs and on obs_next
        -- the NEXT AGENT's observation, quirk Q2 --, GAE, logp_old).  Results are lane batches of one agent each."""
        idx_all, pos, offs = self.aec_partition(buffer)
        results = {}
        for agent, algorithm in self.algorithms.items():
            k = self.agent_idx[agent]
            lo, hi = int(offs[k]), int(offs[k + 1])
            if hi == lo:
                results[agent] = None
                continue
            net = getattr(algorithm, "net", None)
            if net is None or getattr(net, "image_map", None) is None:
                raise NotImplementedError("AEC rows are preprocessed for the fused 64-wide PPO family (DiscreteActorCritic)")
            tind = idx_all[pos[lo:hi]].contiguous()
            n = hi - lo
            obs = buffer._gather(buffer.obs_store, tind).view(n, buffer.obs_dim)
            nxt = buffer.get_device(tind)["obs_next"].view(n, buffer.obs_dim)
            act = buffer._gather(buffer.act_store, tind).view(n)
            P = net.flat.data
            cur = ops.policy_forward(P, obs, net.n_act, net.hidden, image=net.image, mode="given", act=act, want_logits=False)
            v_next = ops.policy_forward(P, nxt, net.n_act, net.hidden, image=net.image, mode="none", want_logits=False)["value"]
            ret, adv = self.aec_returns(buffer, tind, k, cur["value"], v_next, algorithm.gamma, algorithm.gae_lambda, algorithm)
            results[agent] = dict(T=n, rows=None, obs=obs, act=act, v_s=cur["value"], ret=ret.contiguous(), adv=adv.contiguous(),
                                  logp_old=cur["logp"], n_env=1, n_agent=1, aec=True)
        return results

    def dispatch_update_with_batch(self, batch: dict, algorithm_update_with_batch_fn: Callable) -> MapTrainingStats:
        agent_id_to_stats = {}
        for agent_id, algorithm in self.algorithms.items():
            data = batch[agent_id]
            if data is not None and len(data) != 0:
                agent_id_to_stats[agent_id] = algorithm_update_with_batch_fn(algorithm, data, self.agent_idx[agent_id])
        return MapTrainingStats(agent_id_to_stats)


class MultiAgentOnPolicyAlgorithm(nn.Module):
    """marl.py:314-353: each agent's rows update that agent's on-policy algorithm."""

    def __init__(self, *, algorithms: list, env) -> None:
        super().__init__()
        self._dispatcher = MARLDispatcher(algorithms, env)
        self.policy = self._dispatcher.create_policy()
        self._submodules = nn.ModuleList([a for a in _unique(algorithms) if isinstance(a, nn.Module)])

    @property
    def is_within_training_step(self) -> bool:
        return self.policy.is_within_training_step

    @is_within_training_step.setter
    def is_within_training_step(self, v: bool) -> None:
        self.policy.is_within_training_step = v

    def get_algorithm(self, agent_id):
        return self._dispatcher.algorithms[agent_id]

    def _preprocess_batch(self, buffer) -> dict:
        return self._dispatcher.dispatch_process_fn(buffer)

    def _update_with_batch(self, batch: dict, batch_size: int | None, repeat: int, buffer=None) -> MapTrainingStats:
        def update(algorithm, data, agent_col):
            if data.get("aec"):  # one agent's own rows already: no column to select
                return algorithm._update_with_batch(data, batch_size, repeat, agent=None, buffer=None)
            return algorithm._update_with_batch(data, batch_size, repeat, agent=agent_col, buffer=buffer)

        return self._dispatcher.dispatch_update_with_batch(batch, update)

    def update(self, buffer, batch_size: int | None, repeat: int):
        """OnPolicyAlgorithm.update (algorithm_base.py:852-863) through the dispatcher."""
        algos = _unique(self._dispatcher.algorithms.values())
        if len(algos) == 1 and getattr(algos[0], "dispatch", None) == "per_agent" and not getattr(buffer, "aec", False):
            return algos[0].update(buffer, batch_size, repeat)  # shared parameters: the fused/graph path
        if not self.is_within_training_step:
            raise RuntimeError("update() was called outside of a training step as signalled by "
                               "`is_within_training_step=False`")
        for a in algos:
            a.net.sync_image()
        return self._update_with_batch(self._preprocess_batch(buffer), batch_size, repeat, buffer=buffer)

    def state_dict(self, *args, **kwargs):
        return {str(a): alg.state_dict() for a, alg in self._dispatcher.algorithms.items()}

    def load_state_dict(self, sd, *args, **kwargs):
        for a, alg in self._dispatcher.algorithms.items():
            alg.load_state_dict(sd[str(a)])
