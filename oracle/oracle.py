"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy front-end of the CPU oracle (oracle/oracle.c).

May be imported only by tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg.
The product package (tianshou_marl_amd/) never imports this module; its ops fail loudly when
the HIP extension is missing instead of falling back to anything here.

Every function names the reference file:line it restates (paths relative to /root/reference).
Pinned by tests/test_oracle_golden.py against the reference's known-answer tests and the
fixtures under tests/golden/ (generated from the reference itself via oracle/ref_shim.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_vrb_create.restype = C.c_void_p
        _lib.orc_vrb_create.argtypes = [C.c_int64, C.c_int64, C.c_int64]
        for name in ("orc_vrb_maxsize", "orc_vrb_sub_size", "orc_vrb_len"):
            getattr(_lib, name).restype = C.c_int64
            getattr(_lib, name).argtypes = [C.c_void_p]
        _lib.orc_vrb_destroy.argtypes = [C.c_void_p]
        _lib.orc_vrb_reset.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_vrb_add.restype = C.c_int
        _lib.orc_vrb_sample_indices_all.restype = C.c_int64
        _lib.orc_vrb_unfinished_index.restype = C.c_int64
        _lib.orc_vrb_get_buffer_indices.restype = C.c_int64
        _lib.orc_split_bounds.restype = C.c_int64
        for name in ("orc_vrb_last_index", "orc_vrb_lengths"):
            getattr(_lib, name).restype = C.POINTER(C.c_int64)
            getattr(_lib, name).argtypes = [C.c_void_p]
        _lib.orc_vrb_done.restype = C.POINTER(C.c_uint8)
        _lib.orc_vrb_done.argtypes = [C.c_void_p]
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


# ----------------------------------------------------------------------------------------------
# GAE / returns
# ----------------------------------------------------------------------------------------------
def gae(v_s, v_s_next, rew, end_flag, gamma: float, gae_lambda: float) -> np.ndarray:
    """`_gae`, tianshou/algorithm/algorithm_base.py:1079-1134 (flat series, f64)."""
    v_s = np.ascontiguousarray(v_s, np.float64)
    v_s_next = np.ascontiguousarray(v_s_next, np.float64)
    rew = np.ascontiguousarray(rew, np.float64)
    ef = np.ascontiguousarray(np.asarray(end_flag).astype(bool), np.uint8)
    out = np.zeros(rew.shape, np.float64)
    lib().orc_gae(_p(v_s, C.c_double), _p(v_s_next, C.c_double), _p(rew, C.c_double),
                  _p(ef, C.c_uint8), C.c_int64(rew.size), C.c_double(gamma),
                  C.c_double(gae_lambda), _p(out, C.c_double))
    return out


def compute_episodic_return(rew, terminated, truncated, indices, unfinished_index,
                            v_s_next=None, v_s=None, gamma=0.99, gae_lambda=0.95):
    """`Algorithm.compute_episodic_return`, algorithm_base.py:651-717 -> (returns, advantage)."""
    rew = np.ascontiguousarray(rew, np.float64)
    n = rew.size
    term = np.ascontiguousarray(np.asarray(terminated).astype(bool), np.uint8)
    trunc = np.ascontiguousarray(np.asarray(truncated).astype(bool), np.uint8)
    indices = np.ascontiguousarray(indices, np.int64)
    unf = np.ascontiguousarray(unfinished_index, np.int64)
    vn = None if v_s_next is None else np.ascontiguousarray(np.asarray(v_s_next).reshape(-1), np.float64)
    vs = None if v_s is None else np.ascontiguousarray(np.asarray(v_s).reshape(-1), np.float64)
    if v_s_next is None:
        assert np.isclose(gae_lambda, 1.0)  # algorithm_base.py:705
    ret = np.zeros(n, np.float64)
    adv = np.zeros(n, np.float64)
    lib().orc_compute_episodic_return(
        _p(rew, C.c_double), _p(term, C.c_uint8), _p(trunc, C.c_uint8), _p(indices, C.c_int64),
        C.c_int64(n), _p(unf, C.c_int64), C.c_int64(unf.size), _p(vn, C.c_double),
        _p(vs, C.c_double), C.c_double(gamma), C.c_double(gae_lambda), _p(ret, C.c_double),
        _p(adv, C.c_double))
    return ret, adv


def gae_lanes(v_s, v_s_next, rew, terminated, truncated, gamma=0.99, gae_lambda=0.95,
              v_scale: float = 1.0, threads: int = 1):
    """Per-lane GAE on the device layout [T, n_lane] (f32 in, f64 out) -> (returns, adv)."""
    v_s = np.ascontiguousarray(v_s, np.float32)
    T, L = v_s.shape[0], int(np.prod(v_s.shape[1:]))
    v_s_next = np.ascontiguousarray(v_s_next, np.float32)
    rew = np.ascontiguousarray(rew, np.float32)
    term = np.ascontiguousarray(np.asarray(terminated).astype(bool), np.uint8)
    trunc = np.ascontiguousarray(np.asarray(truncated).astype(bool), np.uint8)
    ret = np.zeros(v_s.shape, np.float64)
    adv = np.zeros(v_s.shape, np.float64)
    lib().orc_gae_lanes(_p(v_s, C.c_float), _p(v_s_next, C.c_float), _p(rew, C.c_float),
                        _p(term, C.c_uint8), _p(trunc, C.c_uint8), C.c_int64(T), C.c_int64(L),
                        C.c_double(gamma), C.c_double(gae_lambda), C.c_double(v_scale),
                        _p(ret, C.c_double), _p(adv, C.c_double), C.c_int(threads))
    return ret, adv


def episode_mc_return_to_go(rewards, gamma: float = 0.99) -> np.ndarray:
    """algorithm_base.py:1137-1151."""
    r = np.ascontiguousarray(rewards, np.float64)
    out = np.zeros(r.shape, np.float64)
    lib().orc_episode_mc_return_to_go(_p(r, C.c_double), C.c_int64(r.size), C.c_double(gamma),
                                      _p(out, C.c_double))
    return out


# ----------------------------------------------------------------------------------------------
# VectorReplayBuffer index algebra
# ----------------------------------------------------------------------------------------------
class VectorReplayBufferIndex:
    """Index/episode bookkeeping of `VectorReplayBuffer` (vecbuf.py:15-37, manager.py, buffer_base.py)."""

    def __init__(self, total_size: int, buffer_num: int, rew_dim: int = 1):
        self._l = lib()
        self._h = C.c_void_p(self._l.orc_vrb_create(total_size, buffer_num, rew_dim))
        self.buffer_num = buffer_num
        self.rew_dim = max(1, rew_dim)
        self.maxsize = int(self._l.orc_vrb_maxsize(self._h))
        self.sub_size = int(self._l.orc_vrb_sub_size(self._h))

    def __del__(self):
        try:
            self._l.orc_vrb_destroy(self._h)
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._l.orc_vrb_len(self._h))

    def reset(self, keep_statistics: bool = False) -> None:
        self._l.orc_vrb_reset(self._h, int(keep_statistics))

    def add(self, rew, done, buffer_ids=None):
        """manager.py:131-193 -> (ptr, ep_rew, ep_len, ep_idx)."""
        rew = np.ascontiguousarray(rew, np.float64)
        R = rew.shape[0]
        vector_rew = rew.ndim > 1
        rew2 = rew.reshape(R, -1)
        assert rew2.shape[1] == self.rew_dim
        done = np.ascontiguousarray(np.asarray(done).astype(bool), np.uint8)
        if buffer_ids is None:
            buffer_ids = np.arange(self.buffer_num)
        ids = np.ascontiguousarray(buffer_ids, np.int64)
        ptr = np.zeros(R, np.int64)
        ep_rew = np.zeros((R, self.rew_dim), np.float64)
        ep_len = np.zeros(R, np.int64)
        ep_idx = np.zeros(R, np.int64)
        rc = self._l.orc_vrb_add(self._h, _p(rew2, C.c_double), _p(done, C.c_uint8),
                                 _p(ids, C.c_int64), C.c_int64(R), _p(ptr, C.c_int64),
                                 _p(ep_rew, C.c_double), _p(ep_len, C.c_int64), _p(ep_idx, C.c_int64))
        if rc != 0:
            raise RuntimeError("MalformedBufferError (buffer_base.py:380-386)")
        return ptr, (ep_rew if vector_rew else ep_rew[:, 0]), ep_len, ep_idx

    def sample_indices_all(self) -> np.ndarray:
        out = np.zeros(self.maxsize, np.int64)
        n = self._l.orc_vrb_sample_indices_all(self._h, _p(out, C.c_int64))
        return out[:n].copy()

    def unfinished_index(self) -> np.ndarray:
        out = np.zeros(self.buffer_num, np.int64)
        n = self._l.orc_vrb_unfinished_index(self._h, _p(out, C.c_int64))
        return out[:n].copy()

    def prev(self, index) -> np.ndarray:
        idx = np.ascontiguousarray(np.atleast_1d(index), np.int64)
        out = np.zeros_like(idx)
        self._l.orc_vrb_prev(self._h, _p(idx, C.c_int64), C.c_int64(idx.size), _p(out, C.c_int64))
        return out

    def next(self, index) -> np.ndarray:
        idx = np.ascontiguousarray(np.atleast_1d(index), np.int64)
        out = np.zeros_like(idx)
        self._l.orc_vrb_next(self._h, _p(idx, C.c_int64), C.c_int64(idx.size), _p(out, C.c_int64))
        return out

    def get_buffer_indices(self, start: int, stop: int) -> np.ndarray:
        out = np.zeros(self.maxsize + 1, np.int64)
        n = self._l.orc_vrb_get_buffer_indices(self._h, C.c_int64(start), C.c_int64(stop),
                                               _p(out, C.c_int64))
        if n < 0:
            raise ValueError("Start and stop indices must be within the same subbuffer.")
        return out[:n].copy()

    @property
    def last_index(self) -> np.ndarray:
        return np.ctypeslib.as_array(self._l.orc_vrb_last_index(self._h), (self.buffer_num,)).copy()

    @property
    def lengths(self) -> np.ndarray:
        return np.ctypeslib.as_array(self._l.orc_vrb_lengths(self._h), (self.buffer_num,)).copy()

    @property
    def done(self) -> np.ndarray:
        return np.ctypeslib.as_array(self._l.orc_vrb_done(self._h), (self.maxsize,)).astype(bool)


def split_bounds(length: int, size: int, merge_last: bool = True):
    """`Batch.split` slice bounds, tianshou/data/batch.py:1209-1225."""
    starts = np.zeros(max(1, length), np.int64)
    stops = np.zeros(max(1, length), np.int64)
    n = lib().orc_split_bounds(C.c_int64(length), C.c_int64(size), C.c_int(int(merge_last)),
                               _p(starts, C.c_int64), _p(stops, C.c_int64))
    return [(int(starts[i]), int(stops[i])) for i in range(n)]


# ----------------------------------------------------------------------------------------------
# Agent dispatch (pure numpy: the reference is numpy here too)
# ----------------------------------------------------------------------------------------------
def agent_index(agent_id_rows: np.ndarray, agent) -> np.ndarray:
    """`np.nonzero(batch.obs.agent_id == agent_id)[0]`, multiagent/marl.py:148,233."""
    return np.nonzero(np.asarray(agent_id_rows) == agent)[0]


def dispatch_scatter_act(agent_id_rows: np.ndarray, agents, acts_per_agent) -> np.ndarray:
    """holder.act[agent_index] = act  (marl.py:170-180): merge per-agent actions by row."""
    total = sum(len(a) for a in acts_per_agent)
    first = np.asarray(acts_per_agent[0])
    holder = np.zeros((total, *first.shape[1:]), first.dtype)
    for agent, act in zip(agents, acts_per_agent):
        holder[agent_index(agent_id_rows, agent)] = act
    return holder


# ----------------------------------------------------------------------------------------------
# Categorical + PPO loss
# ----------------------------------------------------------------------------------------------
def categorical_logp_entropy(logits, act):
    """torch Categorical(logits).log_prob/entropy (discrete.py:22-24; ppo.py:160,187,210)."""
    logits = np.ascontiguousarray(logits, np.float32)
    B, A = logits.shape
    act = np.ascontiguousarray(act, np.int64)
    logp = np.zeros(B, np.float64)
    ent = np.zeros(B, np.float64)
    lib().orc_categorical_logp_entropy(_p(logits, C.c_float), _p(act, C.c_int64), C.c_int64(B),
                                       C.c_int64(A), _p(logp, C.c_double), _p(ent, C.c_double))
    return logp, ent


def ppo_loss(logits, act, logp_old, adv, returns, value, v_s_old=None, eps_clip=0.2,
             dual_clip=None, value_clip=False, adv_norm=True, vf_coef=0.5, ent_coef=0.01):
    """`PPO._update_with_batch` loss for one minibatch (ppo.py:182-211) + analytic gradients.

    Returns dict(loss, clip_loss, vf_loss, ent_loss, adv_mean, adv_std, dlogits[M,A], dvalue[M]).
    """
    logits = np.ascontiguousarray(logits, np.float32)
    M, A = logits.shape
    act = np.ascontiguousarray(act, np.int64)
    f = lambda x: None if x is None else np.ascontiguousarray(x, np.float32)  # noqa: E731
    logp_old, adv, returns, value, v_s_old = map(f, (logp_old, adv, returns, value, v_s_old))
    dlogits = np.zeros((M, A), np.float64)
    dvalue = np.zeros(M, np.float64)
    sc = np.zeros(6, np.float64)
    lib().orc_ppo_loss(_p(logits, C.c_float), _p(act, C.c_int64), _p(logp_old, C.c_float),
                       _p(adv, C.c_float), _p(returns, C.c_float), _p(value, C.c_float),
                       _p(v_s_old, C.c_float), C.c_int64(M), C.c_int64(A), C.c_double(eps_clip),
                       C.c_double(dual_clip or 0.0), C.c_int(int(value_clip)),
                       C.c_int(int(adv_norm)), C.c_double(vf_coef), C.c_double(ent_coef),
                       _p(dlogits, C.c_double), _p(dvalue, C.c_double), _p(sc, C.c_double))
    return dict(loss=sc[0], clip_loss=sc[1], vf_loss=sc[2], ent_loss=sc[3], adv_mean=sc[4],
                adv_std=sc[5], dlogits=dlogits, dvalue=dvalue)


def pg_loss(logits, act, adv, returns=None, value=None, vf_coef=0.5, ent_coef=0.01):
    """Plain policy-gradient loss of `A2C._update_with_batch` (a2c.py:260-270) and, with value=None and
    vf_coef = ent_coef = 0, of `Reinforce._update_with_batch` (reinforce.py:373-379; pass adv = returns).

    Returns dict(loss, actor_loss, vf_loss, ent_loss, dlogits[M,A], dvalue[M]).
    """
    logits = np.ascontiguousarray(logits, np.float32)
    M, A = logits.shape
    act = np.ascontiguousarray(act, np.int64)
    f = lambda x: None if x is None else np.ascontiguousarray(x, np.float32)  # noqa: E731
    adv, returns, value = map(f, (adv, returns, value))
    dlogits = np.zeros((M, A), np.float64)
    dvalue = np.zeros(M, np.float64)
    sc = np.zeros(4, np.float64)
    lib().orc_pg_loss(_p(logits, C.c_float), _p(act, C.c_int64), _p(adv, C.c_float), _p(returns, C.c_float),
                      _p(value, C.c_float), C.c_int64(M), C.c_int64(A), C.c_double(vf_coef), C.c_double(ent_coef),
                      _p(dlogits, C.c_double), _p(dvalue, C.c_double), _p(sc, C.c_double))
    return dict(loss=sc[0], actor_loss=sc[1], vf_loss=sc[2], ent_loss=sc[3], dlogits=dlogits, dvalue=dvalue)


class RunningMeanStd:
    """tianshou/utils/statistics.py:68-114 (update only; scalar statistics)."""

    def __init__(self, mean: float = 0.0, std: float = 1.0):
        self._s = np.array([mean, std, 0.0], np.float64)  # NB: reference stores `std` into var (:92)

    def update(self, x) -> None:
        x = np.ascontiguousarray(x, np.float64).reshape(-1)
        lib().orc_rms_update(_p(self._s, C.c_double), _p(x, C.c_double), C.c_int64(x.size))

    mean = property(lambda self: float(self._s[0]))
    var = property(lambda self: float(self._s[1]))
    count = property(lambda self: float(self._s[2]))


# ----------------------------------------------------------------------------------------------
# MLP forward (numpy f64) -- restates tianshou/utils/net/common.py:90-181 (MLP: Linear/ReLU stack,
# flatten(1) at :175-176) for the 2-hidden-layer actor/critic used by the path.
# ----------------------------------------------------------------------------------------------
def mlp_forward(x, weights, biases):
    h = np.asarray(x, np.float64).reshape(len(x), -1)
    n = len(weights)
    for i, (W, b) in enumerate(zip(weights, biases)):
        h = h @ np.asarray(W, np.float64).T + np.asarray(b, np.float64)
        if i + 1 < n:
            h = np.maximum(h, 0.0)
    return h


# ----------------------------------------------------------------------------------------------
# CTDE pieces (numpy): multiagent/ctde.py
# ----------------------------------------------------------------------------------------------
def global_state(obs_by_agent, mode: str = "concatenate") -> np.ndarray:
    """`GlobalStateConstructor.build`, ctde.py:291-300; obs_by_agent ordered by env.agents (Q5)."""
    if mode == "mean":
        return np.stack(obs_by_agent, axis=0).mean(axis=0)
    return np.concatenate(obs_by_agent, axis=-1)


def ctde_td_targets(rew, values, values_next, terminated, gamma: float = 0.99):
    """ctde.py:154-181: multi-output critic -> mean over dim 1; td target; advantage."""
    v = np.asarray(values, np.float64)
    vn = np.asarray(values_next, np.float64)
    if v.ndim > 1 and v.shape[1] > 1:
        v, vn = v.mean(axis=1, keepdims=True), vn.mean(axis=1, keepdims=True)
    elif v.ndim == 1:
        v, vn = v[:, None], vn[:, None]
    rew = np.asarray(rew, np.float64).reshape(len(v), -1)
    term = np.asarray(terminated).astype(bool).reshape(len(v), -1)
    td = rew + gamma * vn * (~term)
    return td, td - v, float(((v - td) ** 2).mean())


def ctde_actor_loss(logp, adv) -> float:
    """ctde.py:184-185: `-(log_probs * advantage).mean()` with log_probs (B,) and advantage (B,1):
    numpy/torch broadcasting makes this a (B,B) outer product (reference quirk Q7)."""
    logp = np.asarray(logp, np.float64).reshape(-1)
    adv = np.asarray(adv, np.float64).reshape(len(logp), -1)
    return float(-(logp * adv).mean())
