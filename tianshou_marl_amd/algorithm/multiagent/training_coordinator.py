"""Multi-agent training coordination: who learns on which step.

Mirror of /root/reference/tianshou/algorithm/multiagent/training_coordinator.py:
  `MATrainer`            :19-263   mode switch, sequential round-robin, checkpoint save/load
  `SimultaneousTrainer`  :266-338  per-agent training frequency, shared-policy handling
  `SequentialTrainer`    :341-410  explicit agent order / steps per agent, train()/eval() toggling
  `SelfPlayTrainer`      :413-571  snapshot pool, uniform / latest / prioritized opponent sampling, win-rate EMA
  `LeaguePlayTrainer`    :574-747  random / elo / win_rate matchmaking, promotion/relegation, Elo update
All of it is host scheduling -- no numerics live here; every `policy.learn(agent_batch)` it issues lands in
the HIP update path (algorithm/ppo.py `learn`, algorithm/multiagent/ctde.py `learn`).  `train_step` takes the
reference's batch format: `batch[agent_id]` = that agent's transitions, optional `global_obs`,
`global_obs_next` at the top level.  `agent_batches_from_buffer` builds that format straight from the device
buffer without leaving HBM.
"""
from __future__ import annotations

import copy
from collections import deque
from typing import Any, Literal

import numpy as np
import torch

from ...data.batch import Batch


def _attach_global(batch: Batch, agent_batch: Batch) -> Batch:
    for k in ("global_obs", "global_obs_next", "chain_done"):
        if k in batch:
            agent_batch[k] = batch[k]
    return agent_batch


class MATrainer:
    VALID_MODES = ["simultaneous", "sequential", "self_play", "league"]

    def __init__(self, policy_manager, training_mode: Literal["simultaneous", "sequential", "self_play", "league"] = "simultaneous",
                 **kwargs: Any) -> None:
        if training_mode not in self.VALID_MODES:
            raise ValueError(f"Invalid training mode: {training_mode}. Must be one of {self.VALID_MODES}")
        self.policy_manager = policy_manager
        self.training_mode = training_mode
        self.step_count = 0
        self.kwargs = kwargs

    # ---- shared helpers ----------------------------------------------------------------------------
    def _policy_of(self, agent_id):
        pm = self.policy_manager
        if getattr(pm, "mode", None) == "shared":
            return pm.policies["shared"]
        return pm.policies.get(agent_id)

    def _learn(self, agent_id, batch: Batch, with_global: bool = True, set_train: bool = False):
        agent_batch = batch[agent_id]
        if with_global:
            _attach_global(batch, agent_batch)
        policy = self._policy_of(agent_id)
        if policy is None:
            return None
        if set_train and hasattr(policy, "train"):
            policy.train()
        return policy.learn(agent_batch)

    def _learn_many(self, agent_ids, batch: Batch, with_global: bool = True) -> dict[str, Any]:
        """`policy.learn(agent_batch)` for several agents / groups.  Data-parallel replicas whose policies share one
        GradSync (parallel.attach_data_parallel(manager, dist)) run their updates in lock step so that the gradients of
        all groups that train this step travel in ONE packed all-reduce per gradient step; otherwise one after the
        other, exactly as the reference does (training_coordinator.py:118,154,336)."""
        pairs = []
        for agent_id in agent_ids:
            policy = self._policy_of(agent_id)
            if policy is None or agent_id not in batch:
                continue
            agent_batch = batch[agent_id]
            if with_global:
                _attach_global(batch, agent_batch)
            pairs.append((agent_id, policy, agent_batch))
        sync = getattr(self.policy_manager, "_grad_sync", None)
        distinct = len({id(p) for _, p, _ in pairs}) == len(pairs)
        if sync is not None and len(pairs) > 1 and distinct and all(
                getattr(p, "_grad_sync", None) is sync and callable(getattr(p, "learn_steps", None)) for _, p, _ in pairs):
            from ...parallel import learn_lockstep, learn_lockstep_graph, lockstep_graphs_enabled

            pairs.sort(key=lambda t: str(t[0]))  # the packing order must not depend on a rank's matchmaking draw
            if lockstep_graphs_enabled() and all(hasattr(p, "_learn_static") and p.learn_graph_ok() for _, p, _ in pairs):
                # the groups' launch sequences and their packed reductions replay from captured graphs (the agreement
                # check on groups AND row counts is the function's own)
                out = learn_lockstep_graph([(p, b, None, 1) for _, p, b in pairs], sync, names=[a for a, _, _ in pairs])
            else:
                sync.check_same([a for a, _, _ in pairs], "the policy groups that train in this step")
                out = learn_lockstep([p.learn_steps(b) for _, p, b in pairs], sync)
            return {a: r for (a, _, _), r in zip(pairs, out)}
        return {a: p.learn(b) for a, p, b in pairs}

    def _init_round_robin(self) -> None:
        if not hasattr(self, "current_agent_idx"):
            self.current_agent_idx = 0
            self.current_agent_steps = 0
            self.agent_order = sorted(self.policy_manager.policies.keys())
            self.steps_per_agent = 1

    def _advance_round_robin(self) -> None:
        self.current_agent_steps += 1
        if self.current_agent_steps >= self.steps_per_agent:
            self.current_agent_steps = 0
            self.current_agent_idx = (self.current_agent_idx + 1) % len(self.agent_order)

    # ---- mode switch (:73-93) ----------------------------------------------------------------------
    def train_step(self, batch: Batch) -> dict[str, Any]:
        self.step_count += 1
        if self.training_mode == "simultaneous":
            return self._simultaneous_train(batch)
        if self.training_mode == "sequential":
            return self._sequential_train(batch)
        if self.training_mode == "self_play":
            return self._self_play_train(batch)
        if self.training_mode == "league":
            return self._league_train(batch)
        raise ValueError(f"Unknown training mode: {self.training_mode}")

    def _simultaneous_train(self, batch: Batch) -> dict[str, Any]:
        return self._learn_many([a for a in self.policy_manager.policies if a in batch], batch)

    def _sequential_train(self, batch: Batch) -> dict[str, Any]:
        self._init_round_robin()
        losses = {}
        current = self.agent_order[self.current_agent_idx]
        if current in batch:
            losses[current] = self.policy_manager.policies[current].learn(_attach_global(batch, batch[current]))
        self._advance_round_robin()
        return losses

    def _self_play_train(self, batch: Batch) -> dict[str, Any]:
        raise NotImplementedError("Use SelfPlayTrainer for self-play training")

    def _league_train(self, batch: Batch) -> dict[str, Any]:
        raise NotImplementedError("Use LeaguePlayTrainer for league play training")

    def set_training_mode(self, mode: str) -> None:
        if mode not in self.VALID_MODES:
            raise ValueError(f"Invalid training mode: {mode}")
        self.training_mode = mode
        if mode == "sequential":
            self._init_round_robin()

    # ---- checkpointing (:205-263) ------------------------------------------------------------------
    def state_dict(self) -> dict[str, Any]:
        return {"step_count": self.step_count, "training_mode": self.training_mode}

    def load_state_dict(self, state: dict[str, Any]) -> None:
        self.step_count = state.get("step_count", 0)
        self.training_mode = state.get("training_mode", "simultaneous")

    def save_checkpoint(self, path: str) -> None:
        policies = {a: p.state_dict() for a, p in self.policy_manager.policies.items() if hasattr(p, "state_dict")}
        torch.save({"trainer_state": self.state_dict(), "policies": policies}, path)

    def load_checkpoint(self, path: str) -> None:
        ckpt = torch.load(path, weights_only=False)
        if "trainer_state" in ckpt:
            self.load_state_dict(ckpt["trainer_state"])
        for agent_id, sd in ckpt.get("policies", {}).items():
            policy = self.policy_manager.policies.get(agent_id)
            if policy is not None and hasattr(policy, "load_state_dict"):
                policy.load_state_dict(sd)


class SimultaneousTrainer(MATrainer):
    def __init__(self, policy_manager, agent_train_freq: dict[str, int] | None = None,
                 replay_buffers: dict | None = None, **kwargs: Any) -> None:
        super().__init__(policy_manager, "simultaneous", **kwargs)
        self.agent_train_freq = agent_train_freq or {}
        self.replay_buffers = replay_buffers or {}

    def train_step(self, batch: Batch) -> dict[str, Any]:
        self.step_count += 1
        pm = self.policy_manager
        agents = pm.agents if getattr(pm, "mode", None) == "shared" else list(pm.policies.keys())
        due = [a for a in agents if self.step_count % self.agent_train_freq.get(a, 1) == 0 and a in batch]
        for agent_id in due:
            policy = self._policy_of(agent_id)
            if policy is not None and hasattr(policy, "train"):
                policy.train()
        return self._learn_many(due, batch, with_global=True)


class SequentialTrainer(MATrainer):
    def __init__(self, policy_manager, agent_order: list[str] | None = None, steps_per_agent: int = 1,
                 **kwargs: Any) -> None:
        super().__init__(policy_manager, "sequential", **kwargs)
        self.agent_order = agent_order if agent_order else sorted(policy_manager.policies.keys())
        self.steps_per_agent = steps_per_agent
        self.current_agent_idx = 0
        self.current_agent_steps = 0

    def train_step(self, batch: Batch) -> dict[str, Any]:
        self.step_count += 1
        losses = {}
        current = self.agent_order[self.current_agent_idx]
        if current in batch:
            for agent_id, p in self.policy_manager.policies.items():  # learner in train mode, the rest in eval
                if hasattr(p, "train"):
                    p.train(agent_id == current)
            losses[current] = self.policy_manager.policies[current].learn(batch[current])
        self._advance_round_robin()
        return losses


class SelfPlayTrainer(MATrainer):
    def __init__(self, policy_manager, main_agent_id: str, snapshot_interval: int = 100, opponent_pool_size: int = 20,
                 opponent_sampling: Literal["uniform", "prioritized", "latest"] = "uniform", win_rate_window: int = 100,
                 **kwargs: Any) -> None:
        super().__init__(policy_manager, "self_play", **kwargs)
        self.main_agent_id = main_agent_id
        self.snapshot_interval = snapshot_interval
        self.opponent_pool_size = opponent_pool_size
        self.opponent_sampling = opponent_sampling
        self.win_rate_window = win_rate_window
        self.opponent_pool: list = []
        self.opponent_win_rates: dict[int, float] = {}
        self.win_history: deque = deque(maxlen=win_rate_window)

    def train_step(self, batch: Batch) -> dict[str, Any]:
        self.step_count += 1
        losses = {}
        if self.main_agent_id in batch:
            policy = self.policy_manager.policies[self.main_agent_id]
            if hasattr(policy, "train"):
                policy.train()
            losses[self.main_agent_id] = policy.learn(batch[self.main_agent_id])
        if self.step_count % self.snapshot_interval == 0:
            self._create_snapshot()
        return losses

    def _create_snapshot(self) -> None:
        """Frozen copy of the learner joins the opponent pool; the oldest one leaves when the pool is full."""
        snapshot = copy.deepcopy(self.policy_manager.policies[self.main_agent_id])
        if hasattr(snapshot, "eval"):
            snapshot.eval()
        self.opponent_pool.append(snapshot)
        if len(self.opponent_pool) > self.opponent_pool_size:
            self.opponent_win_rates.pop(id(self.opponent_pool.pop(0)), None)

    def _sample_opponent(self):
        pool = self.opponent_pool
        if not pool:
            return None
        if self.opponent_sampling == "latest":
            return pool[-1]
        if self.opponent_sampling == "prioritized" and self.opponent_win_rates:
            w = np.array([self.opponent_win_rates.get(id(o), 0.5) + 0.1 for o in pool])  # harder = likelier
            return pool[int(np.random.choice(len(pool), p=w / w.sum()))]
        return pool[int(np.random.choice(len(pool)))]

    def update_win_rate(self, opponent_id: int, won: bool) -> None:
        alpha = 0.1  # exponential moving average, start at 0.5
        prev = self.opponent_win_rates.get(opponent_id, 0.5)
        self.opponent_win_rates[opponent_id] = alpha * (1.0 if won else 0.0) + (1 - alpha) * prev

    def state_dict(self) -> dict[str, Any]:
        """The reference's keys (training_coordinator.py:545-571) plus the opponent pool itself, which the reference leaves
        out (":571 pool persistence" in SURVEY 8f-3): every snapshot's own state_dict, oldest first, and the win rates in
        pool order (the live dict is keyed by object identity, which does not survive a reload)."""
        state = super().state_dict()
        state.update(opponent_pool_size=len(self.opponent_pool), main_agent_id=self.main_agent_id,
                     opponent_win_rates=self.opponent_win_rates,
                     opponent_pool=[p.state_dict() for p in self.opponent_pool if hasattr(p, "state_dict")],
                     opponent_win_rates_by_index=[self.opponent_win_rates.get(id(p), 0.5) for p in self.opponent_pool])
        return state

    def load_state_dict(self, state: dict[str, Any]) -> None:
        super().load_state_dict(state)
        self.main_agent_id = state.get("main_agent_id", self.main_agent_id)
        self.opponent_win_rates = state.get("opponent_win_rates", {})
        saved = state.get("opponent_pool")
        if saved and len(saved) == state.get("opponent_pool_size"):
            # rebuild the snapshots as frozen copies of the learner, then load each one's state
            learner = self.policy_manager.policies[self.main_agent_id]
            self.opponent_pool, self.opponent_win_rates = [], {}
            rates = state.get("opponent_win_rates_by_index", [])
            for i, sd in enumerate(saved):
                snap = copy.deepcopy(learner)
                snap.load_state_dict(sd)
                if hasattr(snap, "eval"):
                    snap.eval()
                self.opponent_pool.append(snap)
                if i < len(rates):
                    self.opponent_win_rates[id(snap)] = float(rates[i])


class LeaguePlayTrainer(MATrainer):
    def __init__(self, policy_manager, league_size: int = 16, promotion_threshold: float = 0.6,
                 relegation_threshold: float = 0.4, matchmaking: Literal["random", "elo", "win_rate"] = "random",
                 games_per_evaluation: int = 10, **kwargs: Any) -> None:
        super().__init__(policy_manager, "league", **kwargs)
        self.league_size = league_size
        self.promotion_threshold = promotion_threshold
        self.relegation_threshold = relegation_threshold
        self.matchmaking = matchmaking
        self.games_per_evaluation = games_per_evaluation
        self.league = list(policy_manager.policies.keys())
        self.agent_performance = {agent: 0.5 for agent in self.league}
        self.elo_ratings = {agent: 1000 for agent in self.league}
        self.game_count = 0
        self.match_history: deque = deque(maxlen=100)

    def train_step(self, batch: Batch) -> dict[str, Any]:
        self.step_count += 1
        self.game_count += 1
        match = [a for a in self._make_match() if a in batch]
        for agent_id in match:
            policy = self.policy_manager.policies[agent_id]
            if hasattr(policy, "train"):
                policy.train()
        # the matched agents learn (training_coordinator.py:641); data-parallel replicas reduce them together
        losses = self._learn_many(match, batch, with_global=False)
        if self.game_count % self.games_per_evaluation == 0:
            self._update_league()
        return losses

    def _make_match(self) -> list[str]:
        if len(self.league) < 2:
            return self.league
        if self.matchmaking in ("elo", "win_rate"):  # neighbours in the rating order play each other
            key = self.elo_ratings if self.matchmaking == "elo" else self.agent_performance
            ranked = sorted(self.league, key=lambda a: key[a])
            i = np.random.randint(0, len(ranked) - 1)
            return [ranked[i], ranked[i + 1]]
        return list(np.random.choice(self.league, size=2, replace=False))

    def _update_league(self) -> tuple[list[str], list[str]]:
        promoted = [a for a, p in self.agent_performance.items() if p >= self.promotion_threshold]
        relegated = [a for a, p in self.agent_performance.items()
                     if p < self.promotion_threshold and p <= self.relegation_threshold]
        return promoted, relegated

    def update_match_result(self, winner: str, loser: str) -> None:
        alpha = 0.1
        self.agent_performance[winner] = alpha + (1 - alpha) * self.agent_performance[winner]
        self.agent_performance[loser] = (1 - alpha) * self.agent_performance[loser]
        self._update_elo(winner, loser)
        self.match_history.append((winner, loser))

    def _update_elo(self, winner: str, loser: str, k: float = 32) -> None:
        rw, rl = self.elo_ratings[winner], self.elo_ratings[loser]
        expected_w = 1 / (1 + 10 ** ((rl - rw) / 400))
        self.elo_ratings[winner] = rw + k * (1 - expected_w)
        self.elo_ratings[loser] = rl + k * (0 - (1 - expected_w))


class DeviceStoreRows:
    """Handle on the rows of a device buffer that `collect(n_step)` left behind (T unrotated, equally filled slots, rows
    chained): the time-major stores themselves, no copies.  Learners that can read the stores in place (CTDEPolicy.learn:
    csrc/critic_train.hip walks the env-major view of the time-major store) take it from `batch.chain_done.store` instead
    of the env-major copies.  `early_done` (device i32[1]) != 0: an episode ended before the last slot, i.e. obs_next of
    that row is not the next slot's obs -- decided on the device, no host round trip."""

    def __init__(self, buffer, T: int, early_done) -> None:
        self.early_done = early_done
        self.T, self.E, self.N, self.D = int(T), buffer.buffer_num, buffer.n_agent, buffer.obs_dim
        self.obs, self.obs_next, self.act = buffer.obs_store, buffer.obs_next_store, buffer.act_store
        self.rew, self.term, self.trunc = buffer.rew_store, buffer.term_store, buffer.trunc_store
        # what the rollout stored beside the rows: logp / V(obs) / V(obs_next) of every column's own policy, and which policy at
        # which parameter version that was (DeviceVectorReplayBuffer.column_outputs) -- None where not stored / unknown
        self.logp, self.vs, self.vnext = buffer.logp_store, buffer.vs_store, buffer.vnext_store
        cols = getattr(buffer, "column_outputs", None)
        self.column_outputs = list(cols) if isinstance(cols, list) else None
        self.key = buffer.storage_key()
        self._buffer = buffer

    def agent_batch(self, a: int) -> Batch:
        """Agent column `a` as env-major copies (the `copies=True` format): for learners that do not read the stores in place."""
        return _agent_rows(self._buffer, a, self.T)


def _agent_rows(buffer, a: int, T: int) -> Batch:
    """Agent column `a` of a uniformly filled device buffer as env-major rows: the agent's fields in ONE launch
    (csrc/gather_fields.hip) instead of six strided torch copies and three dtype conversions -- the same values in the same dtypes;
    the column's stored policy outputs ride along, tagged (see below)."""
    from ... import ops

    E, N = buffer.buffer_num, buffer.n_agent
    n, Dd, dev = E * T, buffer.obs_store.shape[-1], buffer.device
    o = dict(obs=torch.empty(n, Dd, dtype=torch.float32, device=dev), act=torch.empty(n, dtype=torch.int64, device=dev),
             rew=torch.empty(n, dtype=buffer.rew_store.dtype, device=dev), obs_next=torch.empty(n, Dd, dtype=torch.float32, device=dev),
             terminated=torch.empty(n, dtype=torch.bool, device=dev), truncated=torch.empty(n, dtype=torch.bool, device=dev))
    fields = [(buffer.obs_store, o["obs"], T, E, N * Dd, a * Dd), (buffer.act_store, o["act"], T, E, N, a),
              (buffer.rew_store, o["rew"], T, E, N, a), (buffer.obs_next_store, o["obs_next"], T, E, N * Dd, a * Dd),
              (buffer.term_store, o["terminated"], T, E, N, a), (buffer.trunc_store, o["truncated"], T, E, N, a)]
    cols = getattr(buffer, "column_outputs", None)
    if isinstance(cols, list) and cols[a][1] is not None and buffer.vnext_store is not None and buffer.logp_store is not None:
        # the rollout stored this column's log-probabilities, V(obs) and V(obs_next), computed by ITS policy at a known
        # parameter version: they ride along (same launch), tagged, and a learner that is that policy at that version
        # takes them instead of recomputing three forward passes over the rows (PPO.learn; ppo.py:157-161, a2c.py:121-127)
        for key, src_ in (("logp_old", buffer.logp_store), ("v_s", buffer.vs_store), ("v_next", buffer.vnext_store)):
            o[key] = torch.empty(n, dtype=torch.float32, device=dev)
            fields.append((src_, o[key], T, E, N, a))
        o["outputs_policy"], o["outputs_version"] = np.int64(cols[a][0]), np.int64(cols[a][1])
    ops.gather_fields(fields)
    return Batch(**o)


def agent_batches_from_buffer(buffer, agents: list, global_state: bool = True, only: list | None = None,
                              copies: bool = True) -> Batch:
    """The trainers' batch format straight from the device buffer (no host copy).

    Returns Batch({agent: Batch(obs, act, rew, obs_next, terminated, truncated)}, global_obs, global_obs_next) whose
    leaves are HBM tensors.  Rows are every stored joint step in env-major order (flat reference index order,
    `sample_indices(0)`); agent a's rows are column a of the joint rows; the "concatenate" global state of a row
    is that row's `[N*D]` view (GlobalStateConstructor.build, ctde.py:291-294).  `only`: build the batches of these
    agents alone (a league / self-play step trains one agent per team).
    Equally filled sub-buffers that start at slot 0 (what `collect(n_step)` leaves behind a `reset_buffer`) are read as
    strided views of the time-major store -- one copy per field and agent, no index kernel and no host round trip.
    copies=False (chained rows whose episodes end at the last slot only): no env-major copies at all -- every agent's
    batch is `Batch(agent_index=a)` and the stores travel as `batch.chain_done.store` (`DeviceStoreRows`); for learners that
    read the stores in place (CTDEPolicy).  Falls back to the copies when the rows do not qualify.
    copies=False with global_state=False: every agent's batch is `Batch(agent_index=a, store_rows=<handle>)` whose
    `store_rows.store` is the `DeviceStoreRows`: `PPO.learn` gathers the column straight into its static graph buffers (one launch,
    nothing allocated per call) and takes the stored policy outputs where they are its own (`PPO._stored_outputs_ok`)."""
    T = buffer.host_uniform_len()
    if T is not None:
        E = buffer.buffer_num
        em = lambda x: x[:T].transpose(0, 1)  # noqa: E731  [E, T, ...] view: env-major rows without a copy
        d = {"obs": em(buffer.obs_store), "act": em(buffer.act_store), "rew": em(buffer.rew_store),
             "obs_next": em(buffer.obs_next_store) if buffer.obs_next_store is not None else None,
             "terminated": em(buffer.term_store), "truncated": em(buffer.trunc_store)}
        if d["obs_next"] is None:
            d = None
        else:
            col = lambda x, a: x[:, :, a].reshape(E * T, *x.shape[3:])  # noqa: E731  (the reshape is the one copy)
            full = lambda x: x.reshape(E * T, -1)  # noqa: E731
    else:
        d = None
    store_view = d is not None  # `d` holds views of the time-major stores (not rows gathered through get_device)
    if d is None:
        idx = buffer.index.sample_indices_all()
        d = buffer.get_device(idx.cpu().numpy())
        col = lambda x, a: x[:, a].contiguous()  # noqa: E731
        full = lambda x: x.reshape(x.shape[0], -1)  # noqa: E731
    out = Batch()
    store = None
    if global_state and T is not None and T > 0 and buffer.rows_chained is True and buffer.obs_next_store is not None:
        # episodes that end before the last slot break the chain (obs_next of that row is not the next slot's obs): one
        # small device reduction per call (not per agent); the flag stays on the device and the kernels branch on it
        from ... import ops

        early = getattr(buffer, "_early_done_flag", None)  # one allocation per buffer: captured graphs hold its address
        if early is None:
            early = buffer._early_done_flag = torch.zeros(1, dtype=torch.int32, device=buffer.device)
        if T > 1:
            ops.any_nonzero_u8(buffer.done_store[:T - 1].reshape(-1), out=early)
        else:
            early.zero_()
        store = DeviceStoreRows(buffer, T, early)
    if (not copies and not global_state and store_view and T > 0 and buffer.obs_next_store is not None
            and all(getattr(buffer, s_).is_contiguous() for s_ in ("obs_store", "act_store", "rew_store", "obs_next_store", "term_store",
                                                                    "trunc_store"))):
        rows = DeviceStoreRows(buffer, T, None)
        for a, name in enumerate(agents):
            if only is None or name in only:
                handle = buffer.done_store[:0]  # (an empty view: no launch; it only carries the handle through the Batch)
                handle.store = rows
                out[name] = Batch(agent_index=np.int64(a), store_rows=handle)
        return out
    if store is not None and not copies:
        for a, name in enumerate(agents):
            if only is None or name in only:
                out[name] = Batch(agent_index=np.int64(a))
        cd = buffer.done_store[:T].transpose(0, 1).reshape(-1)
        cd.chain_T, cd.store = int(T), store
        out["chain_done"] = cd
        return out
    # (an ignore_obs_next buffer has no obs_next store: its rows come from get_device, which reads obs at next(index))
    fused = (store_view and T > 0 and buffer.obs_next_store is not None
             and all(getattr(buffer, s).is_contiguous() for s in
                     ("obs_store", "act_store", "rew_store", "obs_next_store", "term_store", "trunc_store")))
    for a, name in enumerate(agents):
        if only is not None and name not in only:
            continue
        if fused:
            out[name] = _agent_rows(buffer, a, T)
        else:
            out[name] = Batch(obs=col(d["obs"], a), act=col(d["act"], a).to(torch.int64),
                              rew=col(d["rew"], a), obs_next=col(d["obs_next"], a),
                              terminated=col(d["terminated"], a).bool(), truncated=col(d["truncated"], a).bool())
        if store is not None:
            out[name]["agent_index"] = np.int64(a)
    if global_state:
        out["global_obs"] = full(d["obs"])
        out["global_obs_next"] = full(d["obs_next"])
        if T is not None and T > 0 and buffer.rows_chained is True:
            # rows written by consecutive collects: obs_next of (env, t) is obs of (env, t + 1) unless the episode ended
            # at t.  A learner whose target network IS its online network can then take the values of obs_next from its
            # pass over obs (CTDEPolicy.learn) -- `chain_done` u8 [E * T] (env-major like every other leaf; its attribute
            # `chain_T` = T) says where that does not hold.
            cd = buffer.done_store[:T].transpose(0, 1).reshape(-1)
            cd.chain_T = int(T)
            if store is not None:
                cd.store = store
            out["chain_done"] = cd
    return out
