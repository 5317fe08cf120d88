"""DeviceVectorReplayBuffer -- HBM-resident VectorReplayBuffer for joint-step multi-agent rows.

API mirror of `VectorReplayBuffer(total_size, buffer_num)` / `ReplayBufferManager`
(/root/reference/tianshou/data/buffer/vecbuf.py:15-37, manager.py:13-229, buffer_base.py:173-655):
`add(batch, buffer_ids) -> (ptr, ep_rew, ep_len, ep_idx)`, `sample(0)`, `sample_indices(0)`,
`__getitem__`, `unfinished_index()`, `prev/next`, `reset(keep_statistics)`, `__len__`, `buffer_num`,
`subbuffer_edges`, `get_buffer_indices`, attribute access to stored fields (`buf.rew`, `buf.done` ...).

Storage is time-major SoA in HBM:
    obs[S, B, N, D] f32, obs_next (optional), act[S, B, N] i32, rew[S, B, N] f32,
    terminated/truncated[S, B, N] u8, done[S, B] u8, policy extras logp[S, B, N], v_s[S, B, N]
with S = ceil(total_size / buffer_num) slots and B = buffer_num envs.  One buffer row is one JOINT
step of one env (all N agents), i.e. the parallel-mode layout of EnhancedPettingZooEnv
(enhanced_pettingzoo_env.py:175-222); rew is the reference's per-agent reward vector (R, N).
Flat reference index <-> (env, slot): index = env * S + slot.
All index algebra and payload movement run in HIP (csrc/vrb.hip); there is no host fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .batch import Batch

_RESERVED = ("obs", "act", "rew", "terminated", "truncated", "done", "obs_next", "info", "policy")


class DeviceVectorReplayBuffer:
    def __init__(self, total_size: int, buffer_num: int, n_agent: int, obs_dim: int, device: str | torch.device = "cuda",
                 ignore_obs_next: bool = False, store_policy_outputs: bool = True) -> None:
        self.buffer_num = int(buffer_num)
        self.n_agent = int(n_agent)
        self.obs_dim = int(obs_dim)
        self.device = torch.device(device)
        self.index = ops.VrbState(total_size, buffer_num, rew_dim=n_agent, device=self.device)
        self.sub_size = self.index.sub_size
        self.maxsize = self.index.maxsize
        self._save_obs_next = not ignore_obs_next
        S, B, N, D = self.sub_size, self.buffer_num, self.n_agent, self.obs_dim
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.device)  # noqa: E731
        self.obs_store = z((S, B, N, D), torch.float32)
        self.obs_next_store = z((S, B, N, D), torch.float32) if self._save_obs_next else None
        self.act_store = z((S, B, N), torch.int32)
        self.rew_store = z((S, B, N), torch.float32)
        self.term_store = z((S, B, N), torch.uint8)
        self.trunc_store = z((S, B, N), torch.uint8)
        self.logp_store = z((S, B, N), torch.float32) if store_policy_outputs else None
        self.vs_store = z((S, B, N), torch.float32) if store_policy_outputs else None
        # V(obs_next) per row, filled by the persistent rollout kernel (a2c.py:124 computes it at update time)
        self.vnext_store = z((S, B, N), torch.float32) if store_policy_outputs else None
        # parameter version the stored logp / v_s / v_next of EVERY row were computed with:
        # "empty" (nothing stored yet), an int version, or None (mixed / incomplete -> the update recomputes them)
        self.policy_outputs_version = "empty"
        # the same for logp / v_s alone (rows added by the unfused device collect path carry them but no V(obs_next))
        self.behaviour_outputs_version = "empty"
        self.logp_outputs_version = "empty"  # ... and for logp alone (the actor-only persistent rollout)
        # per AGENT COLUMN (grouped policies: a team's rows carry its own policy's outputs): "empty", None (unknown / mixed), or a
        # list of (policy token, parameter version) -- logp / v_s / v_next of every row of column a were computed by that policy
        # at that version (mark_column_outputs; csrc/rollout_tag.hip)
        self.column_outputs = "empty"
        # "empty" | True | False: every row since the last reset came from collects that continue one another
        # (mark_rows_chained); any add() from elsewhere clears it
        self.rows_chained = "empty"
        self._host_rows: int | None = 0
        self._arange = torch.arange(self.maxsize, dtype=torch.int64, device=self.device)

    def storage_key(self) -> tuple:
        """Identity of the HBM allocations behind this buffer: captured hipGraphs and cached kernel descriptors hold raw
        pointers into them, so they are keyed by this (an `id()` can be recycled by a new object after this one died)."""
        stores = (self.obs_store, self.obs_next_store, self.act_store, self.rew_store, self.term_store, self.trunc_store,
                  self.logp_store, self.vs_store, self.vnext_store, self.index.state, self.index.done_store)
        return tuple(0 if s is None else s.data_ptr() for s in stores)

    # ---- reference attributes ---------------------------------------------------------------
    @property
    def subbuffer_edges(self) -> np.ndarray:
        return np.arange(self.buffer_num + 1, dtype=int) * self.sub_size  # manager.py:55-58

    def __len__(self) -> int:
        return len(self.index)

    @property
    def done_store(self) -> torch.Tensor:
        return self.index.done_store

    def reset(self, keep_statistics: bool = False) -> None:
        self.index.reset(keep_statistics)
        self.policy_outputs_version = "empty"
        self.behaviour_outputs_version = "empty"
        self.logp_outputs_version = "empty"
        self.column_outputs = "empty"
        self.rows_chained = "empty"
        self._host_rows = 0

    # host-side mirror of "every sub-buffer received the same number of rows since reset" (saves the update a
    # device->host round trip just to learn the fill level); None = unknown -> ask the device
    def note_uniform_rows(self, n: int) -> None:
        if self._host_rows is not None:
            self._host_rows += int(n)

    def host_uniform_len(self) -> int | None:
        """Rows per sub-buffer if all are equally filled and start at slot 0, else None."""
        h = self._host_rows
        if h is None or (h > self.sub_size and h % self.sub_size != 0):
            return None
        return min(h, self.sub_size)

    def mark_rows_chained(self, before, chained: bool) -> None:
        """Called by a collect path after its adds.  `rows_chained` is True while EVERY row since the last reset came from
        collects that continue one another: obs_next of slot t is obs of slot t + 1 of the same sub-buffer unless the
        episode ended at t (collector.py:1040-1069) -- what lets the update take V(obs_next) from V(obs) of the next slot
        (ops.value_next_select).  `before` = the marker's value when the collect began (direct add() calls in between
        clear it)."""
        self.rows_chained = bool(chained) and before in ("empty", True)

    def mark_behaviour_outputs(self, version) -> None:
        """Called by a collect path whose every added row carried logp / v_s of the acting policy at `version`
        (None: unknown / not stored)."""
        cur = self.behaviour_outputs_version
        self.behaviour_outputs_version = version if (version is not None and cur in ("empty", version)) else None
        self.mark_logp_outputs(version)

    def mark_column_outputs(self, columns) -> None:
        """Called by a collect path whose every added row carried logp / v_s / v_next of its column's own policy:
        `columns[a]` = (policy token, parameter version); None: not stored / unknown.  Rows of several collects keep the mark only
        while every column stays with one policy at one version."""
        cur = self.column_outputs
        self.column_outputs = list(columns) if (columns is not None and cur in ("empty", list(columns))) else None

    def mark_logp_outputs(self, version) -> None:
        cur = self.logp_outputs_version
        self.logp_outputs_version = version if (version is not None and cur in ("empty", version)) else None

    def mark_policy_outputs(self, version: int) -> None:
        """Called by the fused rollout after it stored logp / v_s / v_next for every row it added."""
        cur = self.policy_outputs_version
        self.policy_outputs_version = version if cur in ("empty", version) else None

    # ---- add --------------------------------------------------------------------------------
    def add_device(self, obs, act, rew, terminated, truncated, obs_next=None, logp=None, v_s=None,
                   buffer_ids=None, done=None, outs=None):
        """Device-path add: every argument is an HBM tensor with leading dim R (rows = envs).

        Returns the reference 4-tuple as device tensors (ptr i64[R], ep_rew f64[R,N], ep_len i64[R], ep_idx i64[R]).
        """
        if done is None:
            done = (terminated.reshape(terminated.shape[0], -1) | truncated.reshape(truncated.shape[0], -1)).any(1)
            # env-level done = any agent done (identical for simple_spread where all agents end together; Q4)
        fields = [(obs, self.obs_store), (act, self.act_store), (rew, self.rew_store),
                  (terminated, self.term_store), (truncated, self.trunc_store)]
        if self._save_obs_next and obs_next is not None:
            fields.append((obs_next, self.obs_next_store))
        if logp is not None and self.logp_store is not None:
            fields.append((logp, self.logp_store))
        if v_s is not None and self.vs_store is not None:
            fields.append((v_s, self.vs_store))
        self.policy_outputs_version = None  # rows added without V(obs_next): the update recomputes critic passes
        self.column_outputs = None
        self.rows_chained = False  # (a Collector that knows its rows continue one another re-marks them afterwards)
        if logp is None or v_s is None or self.logp_store is None:
            self.behaviour_outputs_version = None
        if logp is None or self.logp_store is None:
            self.logp_outputs_version = None
        if buffer_ids is None and rew.shape[0] == self.buffer_num:
            self.note_uniform_rows(1)
        else:
            self._host_rows = None
        return self.index.add(rew, done, buffer_ids, fields=fields, outs=outs)

    def add(self, batch: Batch, buffer_ids=None):
        """Reference signature (manager.py:131-193): host Batch in, numpy 4-tuple out."""
        keys = set(batch.get_keys())
        if not {"obs", "act", "rew", "terminated", "truncated"}.issubset(keys):
            raise ValueError("Input batch must have the keys obs, act, rew, terminated, truncated")
        dev = self.device
        R = len(batch.rew)
        obs = _obs_array(batch.obs)
        t = lambda x, dt: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.array(x))).to(dev, dt).contiguous()  # noqa: E731
        N = self.n_agent
        term = np.broadcast_to(np.asarray(batch.terminated, bool).reshape(R, -1), (R, N))
        trunc = np.broadcast_to(np.asarray(batch.truncated, bool).reshape(R, -1), (R, N))
        rew = np.broadcast_to(np.asarray(batch.rew, np.float32).reshape(R, -1), (R, N))
        obs_next = _obs_array(batch.obs_next) if "obs_next" in keys and self._save_obs_next else None
        ids = None if buffer_ids is None else t(np.asarray(buffer_ids, np.int64), torch.int64)
        pol = batch.get("policy", None)
        logp = t(pol.logp, torch.float32) if isinstance(pol, Batch) and "logp" in pol else None
        v_s = t(pol.v_s, torch.float32) if isinstance(pol, Batch) and "v_s" in pol else None
        out = self.add_device(
            t(obs, torch.float32).reshape(R, N, self.obs_dim), t(np.asarray(batch.act).reshape(R, N), torch.int32),
            t(rew, torch.float32), t(term, torch.uint8), t(trunc, torch.uint8),
            None if obs_next is None else t(obs_next, torch.float32).reshape(R, N, self.obs_dim), logp, v_s, ids)
        ptr, ep_rew, ep_len, ep_idx = (x.cpu().numpy() for x in out)
        return ptr, ep_rew, ep_len, ep_idx

    # ---- index algebra (all on device) ------------------------------------------------------
    def sample_indices(self, batch_size: int | None = 0) -> np.ndarray:
        if batch_size is not None and batch_size < 0:
            return np.array([], int)  # manager.py:197-199
        all_idx = self.index.sample_indices_all()
        if batch_size == 0:
            return all_idx.cpu().numpy()
        n = len(all_idx) if batch_size is None else batch_size
        if len(all_idx) == 0:
            return np.array([], int)
        pick = torch.randint(0, len(all_idx), (n,), device=self.device)
        return all_idx[pick].cpu().numpy()

    def unfinished_index(self) -> np.ndarray:
        return self.index.unfinished_index().cpu().numpy()

    @property
    def last_index(self) -> np.ndarray:
        """Flat index of the row added last in every sub-buffer (manager.py:66, buffer_base.py `last_index`)."""
        return self.index.last_index.cpu().numpy()

    def prev(self, index) -> np.ndarray:
        scalar = np.isscalar(index)
        out = self.index.prev(torch.as_tensor(np.atleast_1d(index), dtype=torch.int64)).cpu().numpy()
        return out[0] if scalar else out

    def next(self, index) -> np.ndarray:
        scalar = np.isscalar(index)
        out = self.index.next(torch.as_tensor(np.atleast_1d(index), dtype=torch.int64)).cpu().numpy()
        return out[0] if scalar else out

    def get_buffer_indices(self, start: int, stop: int) -> np.ndarray:
        """buffer_base.py:173-228 (pure index arithmetic on the sub-buffer edges)."""
        edges = self.subbuffer_edges
        s_edge = int(np.searchsorted(edges, start, side="right")) - 1
        e_edge = int(np.searchsorted(edges, stop - 1, side="right")) - 1
        if s_edge != e_edge:
            raise ValueError(f"Start and stop indices must be within the same subbuffer. Got {start=}, {stop=}.")
        if stop >= start:
            return np.arange(start, stop, dtype=int)
        k_after = int(np.searchsorted(edges, start, side="left"))  # the crossed edge (buffer_base.py:156)
        if k_after == 0:
            raise ValueError(f"The start value should be larger than the first edge, but got {start=}.")
        upper, lower = int(edges[k_after]), int(edges[k_after - 1])
        if lower >= stop:
            raise ValueError(f"The edge before the crossed edge should be smaller than the stop, but got {lower=}, {stop=}.")
        return np.concatenate((np.arange(start, upper, dtype=int), np.arange(lower, stop, dtype=int)))

    # ---- export -----------------------------------------------------------------------------
    def _gather(self, store: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
        return self.index.gather(store, index)

    def get_device(self, index) -> dict[str, torch.Tensor]:
        """Rows by flat reference index as device tensors (ReplayBuffer.__getitem__, buffer_base.py:591-635)."""
        idx = (index if isinstance(index, torch.Tensor) else torch.as_tensor(np.asarray(index))).to(self.device, torch.int64).reshape(-1)
        out = {
            "obs": self._gather(self.obs_store, idx), "act": self._gather(self.act_store, idx),
            "rew": self._gather(self.rew_store, idx), "terminated": self._gather(self.term_store, idx),
            "truncated": self._gather(self.trunc_store, idx),
            "done": self._gather(self.done_store.unsqueeze(-1), idx).squeeze(-1),
        }
        if self._save_obs_next:
            out["obs_next"] = self._gather(self.obs_next_store, idx)
        else:  # ignore_obs_next: obs at next(index) (buffer_base.py:612-616)
            out["obs_next"] = self._gather(self.obs_store, self.index.next(idx))
        if self.logp_store is not None:
            out["logp"] = self._gather(self.logp_store, idx)
            out["v_s"] = self._gather(self.vs_store, idx)
        return out

    def __getitem__(self, index) -> Batch:
        if isinstance(index, slice):
            indices = self.sample_indices(0) if index == slice(None) else np.arange(len(self))[index]
        else:
            indices = index
        d = self.get_device(indices)
        pol = Batch(logp=d.pop("logp").cpu().numpy(), v_s=d.pop("v_s").cpu().numpy()) if "logp" in d else Batch()
        b = Batch({k: v.cpu().numpy() for k, v in d.items()})
        b.rew = b.rew.astype(np.float64)  # the reference stores rew as float (buffer_base.py:484)
        for k in ("terminated", "truncated", "done"):
            b[k] = b[k].astype(bool)
        b.act = b.act.astype(np.int64)
        b.info = Batch()
        b.policy = pol
        return b

    def sample(self, batch_size: int | None = 0) -> tuple[Batch, np.ndarray]:
        indices = self.sample_indices(batch_size)
        return self[indices], indices

    def __getattr__(self, key: str):
        # reference: buffer.rew / buffer.done / ... -> full-length arrays in flat index order
        if key in ("rew", "terminated", "truncated", "done", "obs", "act", "obs_next"):
            b = self.get_device(np.arange(self.maxsize))
            v = b[key].cpu().numpy()
            if key in ("terminated", "truncated", "done"):
                return v.astype(bool)
            return v.astype(np.float64) if key == "rew" else v
        raise AttributeError(key)

    def hasnull(self) -> bool:
        """NaN scan of the stored floats (collector.py:512, trainer.py:928) done on device."""
        bad = torch.isnan(self.obs_store).any() | torch.isnan(self.rew_store).any()
        if self._save_obs_next:
            bad = bad | torch.isnan(self.obs_next_store).any()
        return bool(bad.item())


    def isnull(self) -> Batch:
        """buffer_base.py:649-650: boolean masks (same shapes as `self[:]`) of missing values; computed on device."""
        idx = torch.as_tensor(self.sample_indices(0), dtype=torch.int64, device=self.device)
        d = self.get_device(idx)
        out = Batch()
        for k, v in d.items():
            if k in ("logp", "v_s"):
                continue
            out[k] = (torch.isnan(v) if v.is_floating_point() else torch.zeros_like(v, dtype=torch.bool)).cpu().numpy()
        return out

    _KEY_STORES = {"obs": "obs_store", "obs_next": "obs_next_store", "act": "act_store", "rew": "rew_store",
                   "terminated": "term_store", "truncated": "trunc_store", "done": "done_store"}

    def set_array_at_key(self, seq, key: str, index=None, default_value: float | None = None) -> None:
        """buffer_base.py:637-644 -> Batch.set_array_at_key (batch.py:1269-1305) for the stored (reserved) keys: write
        `seq` at the flat reference indices `index` (None: every slot, len(seq) == maxsize).  New keys cannot be
        created: the vector buffer keeps reserved keys only (manager.py:146-149)."""
        if key not in self._KEY_STORES or getattr(self, self._KEY_STORES[key]) is None:
            raise ValueError(f"Cannot set sequence at key {key}: the device buffer stores {sorted(self._KEY_STORES)} only")
        store = getattr(self, self._KEY_STORES[key])
        idx = np.arange(self.maxsize) if index is None else np.atleast_1d(np.asarray(index))
        if idx.dtype == bool:
            idx = np.nonzero(idx)[0]
        seq = np.asarray(seq)
        if len(seq) != len(idx):
            raise ValueError(f"Length of the sequence ({len(seq)}) must match the number of indices ({len(idx)})")
        idx_d = torch.as_tensor(idx % self.maxsize, dtype=torch.int64, device=self.device)
        env, slot = idx_d // self.sub_size, idx_d % self.sub_size
        store[slot, env] = torch.as_tensor(seq).to(self.device, store.dtype).reshape(len(idx), *store.shape[2:])


class DeviceAECReplayBuffer(DeviceVectorReplayBuffer):
    """The same HBM-resident vector buffer for AEC rows -- the rows `PettingZooEnv` emits and the reference's only runnable
    Collector pipeline stores (pettingzoo_env.py:97-120, SURVEY.md headline fact 6): ONE agent's turn per row,
    `obs = Batch(agent_id, obs[, mask])`, `rew` = the per-agent reward vector, scalar terminated / truncated.
    Per row the buffer keeps the acting agent's index (and the next observation's) next to the payload, so
    `sample(0)` / `__getitem__` hand back the reference's Batch layout (agent ids as strings) while the MARL
    dispatcher partitions rows by agent on the device (`tsm_agent_index`).  Index algebra and episode statistics are the
    shared kernels (csrc/vrb.hip); flat index <-> (env, slot) as for joint rows."""

    aec = True

    def __init__(self, total_size: int, buffer_num: int, agents, obs_dim: int, n_act: int | None = None,
                 device: str | torch.device = "cuda", ignore_obs_next: bool = False) -> None:
        super().__init__(total_size, buffer_num, 1, obs_dim, device=device, ignore_obs_next=ignore_obs_next,
                         store_policy_outputs=False)
        self.agents = list(agents)
        self.agent_idx = {a: i for i, a in enumerate(self.agents)}
        self.rew_dim = len(self.agents)
        S, B = self.sub_size, self.buffer_num
        self.index = ops.VrbState(total_size, buffer_num, rew_dim=self.rew_dim, device=self.device)
        self.rew_store = torch.zeros(S, B, self.rew_dim, dtype=torch.float32, device=self.device)
        self.agent_store = torch.zeros(S, B, dtype=torch.int32, device=self.device)
        self.agent_next_store = torch.zeros(S, B, dtype=torch.int32, device=self.device) if self._save_obs_next else None
        self.n_act = n_act
        self.mask_store = torch.ones(S, B, n_act, dtype=torch.uint8, device=self.device) if n_act else None
        self._names = np.array(self.agents, dtype=object)

    def storage_key(self) -> tuple:
        return super().storage_key() + (self.agent_store.data_ptr(),)

    def _codes(self, agent_id) -> torch.Tensor:
        ids = np.asarray(agent_id)
        codes = np.empty(ids.shape, np.int32)
        for i, a in enumerate(ids.reshape(-1)):
            if a not in self.agent_idx:
                raise ValueError(f"unknown agent id {a!r} (buffer was built for {self.agents})")
            codes.reshape(-1)[i] = self.agent_idx[a]
        return torch.as_tensor(codes).to(self.device)

    def add(self, batch: Batch, buffer_ids=None):
        """manager.py:131-193 for AEC rows: host Batch in (obs / obs_next = Batch(agent_id, obs[, mask])), numpy 4-tuple out."""
        keys = set(batch.get_keys())
        if not {"obs", "act", "rew", "terminated", "truncated"}.issubset(keys):
            raise ValueError("Input batch must have the keys obs, act, rew, terminated, truncated")
        obs = batch.obs
        if not (isinstance(obs, Batch) and "agent_id" in obs and "obs" in obs):
            raise ValueError("AEC rows carry obs = Batch(agent_id, obs[, mask]) (pettingzoo_env.py:76-93)")
        dev = self.device
        R = len(np.asarray(batch.terminated).reshape(-1))
        t = lambda x, dt: torch.as_tensor(np.ascontiguousarray(x)).to(dev, dt).contiguous()  # noqa: E731
        rew = np.asarray(batch.rew, np.float32).reshape(R, -1)
        if rew.shape[1] != self.rew_dim:
            raise ValueError(f"rew has {rew.shape[1]} entries per row, the env has {self.rew_dim} agents")
        term_h = np.asarray(batch.terminated, bool).reshape(R)
        trunc_h = np.asarray(batch.truncated, bool).reshape(R)
        term, trunc = t(term_h.reshape(R, 1), torch.uint8), t(trunc_h.reshape(R, 1), torch.uint8)
        fields = [(t(np.asarray(obs.obs, np.float32).reshape(R, 1, self.obs_dim), torch.float32), self.obs_store),
                  (t(np.asarray(batch.act).reshape(R, 1), torch.int32), self.act_store),
                  (t(rew, torch.float32), self.rew_store), (term, self.term_store), (trunc, self.trunc_store),
                  (self._codes(obs.agent_id).reshape(R), self.agent_store)]
        if self.mask_store is not None and "mask" in obs:
            fields.append((t(np.asarray(obs.mask, bool).reshape(R, self.n_act), torch.uint8), self.mask_store))
        if self._save_obs_next and "obs_next" in keys:
            nxt = batch.obs_next
            fields.append((t(np.asarray(nxt.obs, np.float32).reshape(R, 1, self.obs_dim), torch.float32), self.obs_next_store))
            fields.append((self._codes(nxt.agent_id).reshape(R), self.agent_next_store))
        ids = None if buffer_ids is None else t(np.asarray(buffer_ids, np.int64), torch.int64)
        if ids is None and R == self.buffer_num:
            self.note_uniform_rows(1)
        else:
            self._host_rows = None
        done = t(term_h | trunc_h, torch.uint8)
        self.rows_chained = False  # agent turns: the next row of a sub-buffer belongs to another agent
        out = self.index.add(t(rew, torch.float32), done, ids, fields=fields)
        return tuple(x.cpu().numpy() for x in out)

    def add_device(self, *args, **kwargs):
        raise NotImplementedError("AEC rows are added through add(batch, buffer_ids) (host Collector loop)")

    def get_device(self, index) -> dict[str, torch.Tensor]:
        idx = (index if isinstance(index, torch.Tensor) else torch.as_tensor(np.asarray(index))).to(self.device, torch.int64).reshape(-1)
        out = super().get_device(idx)
        out["agent"] = self._gather(self.agent_store.unsqueeze(-1), idx).view(-1)
        if self._save_obs_next:
            out["agent_next"] = self._gather(self.agent_next_store.unsqueeze(-1), idx).view(-1)
        else:
            out["agent_next"] = self._gather(self.agent_store.unsqueeze(-1), self.index.next(idx)).view(-1)
        if self.mask_store is not None:
            out["mask"] = self._gather(self.mask_store, idx)
        return out

    def __getitem__(self, index) -> Batch:
        if isinstance(index, slice):
            indices = self.sample_indices(0) if index == slice(None) else np.arange(len(self))[index]
        else:
            indices = index
        d = {k: v.cpu().numpy() for k, v in self.get_device(indices).items()}
        n = len(d["agent"])
        obs = Batch(agent_id=self._names[d["agent"]], obs=d["obs"].reshape(n, self.obs_dim))
        obs_next = Batch(agent_id=self._names[d["agent_next"]], obs=d["obs_next"].reshape(n, self.obs_dim))
        if "mask" in d:
            obs.mask = d["mask"].astype(bool)
        b = Batch(obs=obs, act=d["act"].reshape(n).astype(np.int64), rew=d["rew"].astype(np.float64),
                  terminated=d["terminated"].reshape(n).astype(bool), truncated=d["truncated"].reshape(n).astype(bool),
                  done=d["done"].astype(bool), obs_next=obs_next)
        b.info = Batch()
        b.policy = Batch()
        return b


class VectorReplayBuffer(DeviceVectorReplayBuffer):
    """`VectorReplayBuffer(total_size, buffer_num)` with the reference's constructor (vecbuf.py:15-37): like the
    reference (manager.py:187-192) the stores are allocated lazily -- here when a Collector binds the buffer to its env
    (agent count, observation width) or at the first `add`.  Everything else is DeviceVectorReplayBuffer."""

    def __init__(self, total_size: int, buffer_num: int, n_agent: int | None = None, obs_dim: int | None = None,
                 device: str | torch.device = "cuda", **kwargs) -> None:
        self._lazy = dict(total_size=int(total_size), buffer_num=int(buffer_num), device=device, kwargs=kwargs)
        self.buffer_num = int(buffer_num)
        self.sub_size = -(-int(total_size) // self.buffer_num)
        self.maxsize = self.sub_size * self.buffer_num
        if n_agent is not None and obs_dim is not None:
            self.bind(n_agent, obs_dim)

    @property
    def allocated(self) -> bool:
        return "index" in self.__dict__

    def bind(self, n_agent: int, obs_dim: int) -> "VectorReplayBuffer":
        if self.allocated:
            if (self.n_agent, self.obs_dim) != (int(n_agent), int(obs_dim)):
                raise ValueError(f"buffer holds rows of {self.n_agent} agents x {self.obs_dim} floats, "
                                 f"asked to bind {n_agent} x {obs_dim}")
            return self
        lz = self._lazy
        DeviceVectorReplayBuffer.__init__(self, lz["total_size"], lz["buffer_num"], int(n_agent), int(obs_dim),
                                          device=lz["device"], **lz["kwargs"])
        return self

    def add(self, batch: Batch, buffer_ids=None):
        if not self.allocated:
            obs = _obs_array(batch.obs)
            obs = obs.reshape(len(batch.rew), -1, obs.shape[-1]) if obs.ndim > 2 else obs.reshape(len(batch.rew), 1, -1)
            self.bind(obs.shape[1], obs.shape[2])
        return super().add(batch, buffer_ids)

    def __len__(self) -> int:
        return super().__len__() if self.allocated else 0

    def reset(self, keep_statistics: bool = False) -> None:
        if self.allocated:
            super().reset(keep_statistics)

    def __getattr__(self, key: str):
        if key in ("index", "obs_store") or key.endswith("_store"):
            raise AttributeError(f"{key}: the buffer is not allocated yet (bind it to an env through a Collector, or add a row)")
        return super().__getattr__(key)


def _obs_array(obs) -> np.ndarray:
    """Accept the reference's observation containers: plain array, Batch(obs=...), or the parallel-mode
    Batch(observations=Batch(agent_i=...)) (enhanced_pettingzoo_env.py:202-205) -> [R, N, D] in agent order."""
    if isinstance(obs, Batch):
        if "observations" in obs:
            o = obs.observations
            names = list(o.get_keys())
            if "agent_ids" in obs and len(np.asarray(obs.agent_ids)) > 0:
                first = np.asarray(obs.agent_ids)
                order = list(first[0]) if first.ndim > 1 else list(first)
                names = [n for n in order if n in o]  # env.agents order, never dict order (quirk Q5)
            return np.stack([np.asarray(o[n], np.float32) for n in names], axis=1)
        if "obs" in obs:
            return np.asarray(obs.obs, np.float32)
        raise ValueError("unsupported observation Batch")
    return np.asarray(obs, np.float32)
