#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported through oracle/ref_shim.py).

Test infrastructure.  Needs /root/reference (not present on the GPU box): run it in the build
container only; the resulting small .npz fixtures are committed, the reference never travels.

    python tests/golden/make_fixtures.py            # regenerate everything

Each fixture holds inputs + the reference's outputs for one hot-path function
(SURVEY.md section 8a rows in brackets):
  gae.npz          Algorithm.compute_episodic_return / _gae            [a12]
  vrb_trace.npz    VectorReplayBuffer.add / sample_indices(0) / prev / next / unfinished_index /
                   get_buffer_indices / reset                           [a8, a9]
  pg_update.npz    A2C / Reinforce _preprocess_batch + _update_with_batch (returns, loss statistics, post-update
                   weights, last gradients)                             [(f)4]
  ppo_update.npz   PPO._preprocess_batch + PPO._update_with_batch (loss scalars, gradients,
                   Adam-updated weights, minibatch permutation)         [a7, a11, a13, a14, a15]
  ppo_update_wide.npz  the same for Net(hidden_sizes=[128, 128]) on 48-wide observations  [a7, a14 at configs[2]'s widths]
  marl_dispatch.npz MultiAgentPolicy.forward scatter + MARLDispatcher per-agent GAE (incl. quirk Q1),
                   FlexibleMultiAgentPolicyManager shared forward       [a5, a6, a10]
  ctde.npz         GlobalStateConstructor.build + CTDEPolicy.learn      [a16]
  ctde_wide.npz    CTDEPolicy.learn with a 128-wide actor and centralized critic at N = 4, D = 24 (critic input 96) on rows
                   laid out as a time-major store: every agent's learn() in turn, with and without episodes ending mid-store
                   [a16, a17]
  ctde_c3.npz      the same at configs[2]'s own widths: N = 8, D = 48 (critic input 384, 8 outputs)   [a16, a17]
  async_collector.npz  AsyncCollector over an async vector env with scripted readiness: statistics, ready sets, buffer rows  [(f)4]
  collector.npz    the synchronous Collector through ten scripted collect / reset calls (truncating envs, surplus-env removal,
                   reset_before_collect / reset_buffer / reset_stat): statistics, counters, next observations, buffer rows   [a4]
  collector_port.npz  the Collector in the one mode the CPU baseline port restates (oracle/cpu_path.py::PortCollector: n_step calls,
                   reset_buffer(keep_statistics=True) between them): statistics and every buffer row per call   [a4, SURVEY 8d(iv)]
  trainers.npz     the training coordinators as a trace under seeded numpy randomness: who learns, which snapshot is sampled,
                   every Elo / performance / win rate, promotion and relegation lists   [a17]
  misc.npz         Batch.split bounds, RunningMeanStd, episode_mc_return_to_go   [a11, a14]
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_shim  # noqa: E402

ref_shim.install()

import logging  # noqa: E402

logging.getLogger("tianshou.data.buffer.buffer_base").setLevel(logging.CRITICAL)

import torch  # noqa: E402
from tianshou.algorithm import Algorithm  # noqa: E402
from tianshou.algorithm.algorithm_base import episode_mc_return_to_go  # noqa: E402
from tianshou.algorithm.modelfree.ppo import PPO  # noqa: E402
from tianshou.algorithm.modelfree.reinforce import DiscreteActorPolicy  # noqa: E402
from tianshou.algorithm.multiagent.ctde import (  # noqa: E402
    CentralizedCritic,
    CTDEPolicy,
    DecentralizedActor,
    GlobalStateConstructor,
)
from tianshou.algorithm.multiagent.flexible_policy import FlexibleMultiAgentPolicyManager  # noqa: E402
from tianshou.algorithm.multiagent.marl import MultiAgentOnPolicyAlgorithm, MultiAgentPolicy  # noqa: E402
from tianshou.algorithm.algorithm_base import Policy  # noqa: E402
from tianshou.algorithm.optim import AdamOptimizerFactory  # noqa: E402
from tianshou.data import Batch, ReplayBuffer, VectorReplayBuffer  # noqa: E402
from tianshou.utils import RunningMeanStd  # noqa: E402
from tianshou.utils.net.common import Net  # noqa: E402
from tianshou.utils.net.discrete import DiscreteActor, DiscreteCritic  # noqa: E402
from tianshou.utils.torch_utils import policy_within_training_step  # noqa: E402
import gymnasium as gym  # noqa: E402  (the shim's fake)


def save(name: str, **arrays) -> None:
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {len(arrays)} arrays, {os.path.getsize(path)} bytes")


# ------------------------------------------------------------------------------------------------
def fill_vector_buffer(rng, n_env, T, obs_dim, p_term=0.05, trunc_at=None, rew_dim=0, n_act=5):
    """Fill a reference VectorReplayBuffer with T vector steps of synthetic rows."""
    buf = VectorReplayBuffer(n_env * T, n_env)
    obs = rng.standard_normal((n_env, obs_dim)).astype(np.float32)
    for t in range(T):
        obs_next = rng.standard_normal((n_env, obs_dim)).astype(np.float32)
        rew = rng.standard_normal((n_env, rew_dim) if rew_dim else (n_env,))
        term = rng.random(n_env) < p_term
        trunc = np.zeros(n_env, bool)
        if trunc_at is not None and (t + 1) % trunc_at == 0:
            trunc[:] = True
        b = Batch(obs=obs, act=rng.integers(0, n_act, n_env), rew=rew, terminated=term,
                  truncated=trunc, obs_next=obs_next, info=Batch(env_id=np.arange(n_env)))
        buf.add(b, buffer_ids=np.arange(n_env))
        obs = obs_next
    return buf


def make_gae() -> None:
    rng = np.random.default_rng(0)
    out = {}
    cases = [
        dict(n_env=4, T=25, p_term=0.05, trunc_at=None, gamma=0.99, lam=0.95),
        dict(n_env=3, T=10, p_term=0.2, trunc_at=5, gamma=0.9, lam=0.8),
        dict(n_env=16, T=25, p_term=0.01, trunc_at=25, gamma=0.99, lam=0.95),
        dict(n_env=2, T=7, p_term=0.0, trunc_at=None, gamma=0.99, lam=1.0),
    ]
    for ci, c in enumerate(cases):
        buf = fill_vector_buffer(rng, c["n_env"], c["T"], 3, c["p_term"], c["trunc_at"])
        batch, indices = buf.sample(0)
        n = len(indices)
        v_s = rng.standard_normal(n).astype(np.float32)
        v_s_ = rng.standard_normal(n).astype(np.float32)
        ret, adv = Algorithm.compute_episodic_return(
            batch, buf, indices, torch.from_numpy(v_s_), torch.from_numpy(v_s),
            gamma=c["gamma"], gae_lambda=c["lam"])
        p = f"c{ci}_"
        out.update({
            p + "rew": batch.rew, p + "terminated": batch.terminated, p + "truncated": batch.truncated,
            p + "indices": indices, p + "unfinished": buf.unfinished_index(), p + "v_s": v_s,
            p + "v_s_next": v_s_, p + "gamma": c["gamma"], p + "lam": c["lam"],
            p + "returns": ret, p + "adv": adv, p + "n_env": c["n_env"], p + "T": c["T"],
        })
        assert ret.dtype == np.float64
    # v_s=None (roll) and v_s_=None (MC return, gae_lambda=1) branches
    buf = fill_vector_buffer(rng, 3, 9, 3, 0.15, None)
    batch, indices = buf.sample(0)
    v_s_ = rng.standard_normal(len(indices))
    ret, adv = Algorithm.compute_episodic_return(batch, buf, indices, v_s_, None, gamma=0.97, gae_lambda=0.9)
    out.update(roll_rew=batch.rew, roll_terminated=batch.terminated, roll_truncated=batch.truncated,
               roll_indices=indices, roll_unfinished=buf.unfinished_index(), roll_v_s_next=v_s_,
               roll_returns=ret, roll_adv=adv)
    ret, adv = Algorithm.compute_episodic_return(batch, buf, indices, None, None, gamma=0.97, gae_lambda=1.0)
    out.update(mc_returns=ret, mc_adv=adv)
    save("gae.npz", n_cases=len(cases), **out)


# ------------------------------------------------------------------------------------------------
def make_vrb_trace() -> None:
    rng = np.random.default_rng(1)
    out = {}
    configs = [(20, 4, 0), (23, 5, 0), (12, 3, 4), (64, 8, 3)]  # (total_size, buffer_num, rew_dim)
    for ci, (total, num, rew_dim) in enumerate(configs):
        buf = VectorReplayBuffer(total, num)
        p = f"c{ci}_"
        out[p + "cfg"] = np.array([total, num, rew_dim])
        n_add = 40
        out[p + "n_add"] = n_add
        for ai in range(n_add):
            if ai == 25:  # exercise reset(keep_statistics=True) mid-trace (trainer.py:1104)
                buf.reset(keep_statistics=True)
                out[p + "reset_at"] = ai
            k = int(rng.integers(1, num + 1))
            ids = np.sort(rng.choice(num, size=k, replace=False))
            rew = rng.integers(-3, 4, size=(k, rew_dim) if rew_dim else (k,)).astype(np.float64)
            term = rng.random(k) < 0.2
            trunc = rng.random(k) < 0.1
            b = Batch(obs=np.zeros(k), act=np.zeros(k, int), rew=rew, terminated=term, truncated=trunc)
            ptr, ep_rew, ep_len, ep_idx = buf.add(b, buffer_ids=ids)
            q = f"{p}a{ai}_"
            out.update({q + "ids": ids, q + "rew": rew, q + "term": term, q + "trunc": trunc,
                        q + "ptr": ptr, q + "ep_rew": np.asarray(ep_rew, np.float64),
                        q + "ep_len": ep_len, q + "ep_idx": ep_idx,
                        q + "len": len(buf), q + "unfinished": buf.unfinished_index(),
                        q + "sample0": buf.sample_indices(0)})
            if ai % 5 == 4:
                allidx = np.arange(-2, buf.maxsize + 2)
                out[q + "prev"] = buf.prev(allidx)
                out[q + "next"] = buf.next(allidx)
                out[q + "done"] = np.array(buf.done, copy=True)
                out[q + "last_index"] = np.array(buf.last_index, copy=True)
        # get_buffer_indices (buffer_base.py:173-228), incl. wrap-around and errors
        gbi = []
        S = buf.maxsize // num
        for start in range(0, buf.maxsize):
            for stop in range(0, buf.maxsize + 1):
                try:
                    r = buf.get_buffer_indices(start, stop)
                    gbi.append((start, stop, len(r), int(r.sum()) if len(r) else 0,
                                int(r[0]) if len(r) else -1, int(r[-1]) if len(r) else -1))
                except ValueError:
                    gbi.append((start, stop, -1, 0, -1, -1))
        out[p + "gbi"] = np.array(gbi, np.int64)
        out[p + "sub_size"] = S
    save("vrb_trace.npz", n_cfg=len(configs), **out)


# ------------------------------------------------------------------------------------------------
def build_ppo(obs_dim, n_act, hidden, seed, lr=3e-4, **ppo_kw):
    torch.manual_seed(seed)
    actor = DiscreteActor(preprocess_net=Net(state_shape=(obs_dim,), hidden_sizes=hidden),
                          action_shape=n_act, softmax_output=False)
    critic = DiscreteCritic(preprocess_net=Net(state_shape=(obs_dim,), hidden_sizes=hidden))
    for m in list(actor.modules()) + list(critic.modules()):
        if isinstance(m, torch.nn.Linear):  # test/discrete/test_ppo_discrete.py:103-106
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    policy = DiscreteActorPolicy(actor=actor, action_space=gym.spaces.Discrete(n_act))
    algo = PPO(policy=policy, critic=critic, optim=AdamOptimizerFactory(lr=lr), **ppo_kw)
    return algo, actor, critic


def net_params(actor, critic):
    lin = lambda m: [x for x in m.modules() if isinstance(x, torch.nn.Linear)]  # noqa: E731
    d = {}
    for name, mod in (("actor", actor), ("critic", critic)):
        for i, l in enumerate(lin(mod)):
            d[f"{name}_w{i}"] = l.weight.detach().numpy().copy()
            d[f"{name}_b{i}"] = l.bias.detach().numpy().copy()
    return d


def net_grads(actor, critic):
    lin = lambda m: [x for x in m.modules() if isinstance(x, torch.nn.Linear)]  # noqa: E731
    d = {}
    for name, mod in (("actor", actor), ("critic", critic)):
        for i, l in enumerate(lin(mod)):
            d[f"{name}_gw{i}"] = l.weight.grad.detach().numpy().copy()
            d[f"{name}_gb{i}"] = l.bias.grad.detach().numpy().copy()
    return d


def _ppo_update_variants(variants, n_env, T, obs_dim, n_act, hidden) -> dict:
    """Run the reference's `_preprocess_batch` + `_update_with_batch` for every variant; inputs and outputs by key."""
    out = {}
    for v in variants:
        rng = np.random.default_rng(7)
        p = v["name"] + "_"
        algo, actor, critic = build_ppo(obs_dim, n_act, hidden, seed=3, **v["ppo"])
        buf = fill_vector_buffer(rng, n_env, T, obs_dim, p_term=0.03, trunc_at=25, n_act=n_act)
        batch, indices = buf.sample(0)
        out.update({p + k: val for k, val in net_params(actor, critic).items()})
        out.update({p + "obs": batch.obs, p + "obs_next": batch.obs_next, p + "act": batch.act,
                    p + "rew": batch.rew, p + "terminated": batch.terminated,
                    p + "truncated": batch.truncated, p + "indices": indices,
                    p + "unfinished": buf.unfinished_index()})
        if v["ppo"].get("return_scaling"):
            algo.ret_rms.update(rng.standard_normal(50) * 3.0)  # non-trivial running stats
            out[p + "rms_before"] = np.array([algo.ret_rms.mean, algo.ret_rms.var, algo.ret_rms.count], np.float64)
        with policy_within_training_step(algo.policy):
            pb = algo._preprocess_batch(batch, buf, indices)
            out.update({p + "v_s": pb.v_s.numpy().copy(), p + "returns": pb.returns.numpy().copy(),
                        p + "adv": pb.adv.numpy().copy(), p + "logp_old": pb.logp_old.numpy().copy()})
            if v["ppo"].get("return_scaling"):
                out[p + "rms_after"] = np.array([algo.ret_rms.mean, algo.ret_rms.var, algo.ret_rms.count], np.float64)
            # logits of the pre-update policy for the loss-kernel fixture
            with torch.no_grad():
                lg, _ = actor(torch.from_numpy(batch.obs))
                out[p + "logits"] = lg.numpy().copy()
            # the permutation Batch.split will draw (batch.py:1219): replay np.random state
            np.random.seed(11)
            n = len(indices)
            perms = []
            st = np.random.get_state()
            for _ in range(v["repeat"]):
                perms.append(np.random.permutation(n))
            np.random.set_state(st)
            out[p + "perms"] = np.stack(perms)
            with torch.enable_grad():
                algo.train()
                stats = algo._update_with_batch(pb, v["batch_size"], v["repeat"])
        if v["ppo"].get("return_scaling"):  # (recompute_advantage updates the statistics again before every later repeat)
            out[p + "rms_final"] = np.array([algo.ret_rms.mean, algo.ret_rms.var, algo.ret_rms.count], np.float64)
        out[p + "batch_size"] = -1 if v["batch_size"] is None else v["batch_size"]
        out[p + "repeat"] = v["repeat"]
        out[p + "recompute_advantage"] = int(bool(v["ppo"].get("recompute_advantage")))
        out[p + "gradient_steps"] = stats.gradient_steps
        # per-gradient-step loss values are summarised by SequenceSummaryStats; keep mean/std/min/max
        for k in ("loss", "actor_loss", "vf_loss", "ent_loss"):
            s = getattr(stats, k)
            out[p + "stat_" + k] = np.array([s.mean, s.std, s.max, s.min], np.float64)
        out.update({p + "after_" + k: val for k, val in net_params(actor, critic).items()})
        out.update({p + "last_" + k: val for k, val in net_grads(actor, critic).items()})
        out[p + "ppo_cfg"] = np.array([algo.eps_clip, algo.dual_clip or 0.0, float(algo.value_clip),
                                       float(algo.advantage_normalization), algo.vf_coef, algo.ent_coef,
                                       algo.gamma, algo.gae_lambda,
                                       v["ppo"].get("max_grad_norm") or 0.0, 3e-4], np.float64)
    return out


def make_ppo_update() -> None:
    variants = [
        dict(name="default", ppo={}, batch_size=None, repeat=1),
        dict(name="mb64", ppo={}, batch_size=64, repeat=2),
        dict(name="dualclip_vclip", ppo=dict(dual_clip=2.0, value_clip=True, max_grad_norm=0.5),
             batch_size=100, repeat=1),
        dict(name="noadvnorm_retscale", ppo=dict(advantage_normalization=False, return_scaling=True),
             batch_size=None, repeat=1),
        # ppo.py:174-178: returns / advantages recomputed with the current critic before every repeat after the first
        # (logp_old stays); with return_scaling the running statistics are updated again by every recomputation (a2c.py:132-146)
        dict(name="recompute", ppo=dict(recompute_advantage=True), batch_size=64, repeat=2),
        dict(name="recompute_retscale_vclip", ppo=dict(recompute_advantage=True, return_scaling=True, value_clip=True),
             batch_size=100, repeat=3),
    ]
    out = _ppo_update_variants(variants, n_env=8, T=25, obs_dim=18, n_act=5, hidden=[64, 64])
    save("ppo_update.npz", variants=np.array([v["name"] for v in variants]), **out)


def make_ppo_update_wide() -> None:
    """The same runs for `Net(hidden_sizes=[128, 128])` on 48-wide observations (the actor / local-critic shape of BASELINE
    configs[2]): pins the 128-wide kernels (csrc/ppo_rows.hip, csrc/critic_rows.hip) to the reference's own tensors."""
    variants = [
        dict(name="w128_mb64", ppo={}, batch_size=64, repeat=2),
        dict(name="w128_vclip_gn", ppo=dict(value_clip=True, max_grad_norm=0.5), batch_size=64, repeat=2),
        dict(name="w128_dualclip_full", ppo=dict(dual_clip=2.0, value_clip=True), batch_size=None, repeat=1),
        dict(name="w128_recompute", ppo=dict(recompute_advantage=True), batch_size=64, repeat=2),
    ]
    out = _ppo_update_variants(variants, n_env=8, T=25, obs_dim=48, n_act=5, hidden=[128, 128])
    save("ppo_update_wide.npz", variants=np.array([v["name"] for v in variants]), **out)


def make_pg_update() -> None:
    """A2C._update_with_batch (a2c.py:247-285) and Reinforce._preprocess_batch/_update_with_batch
    (reinforce.py:273-311, 361-379) on the same buffer contents as ppo_update.npz."""
    from tianshou.algorithm.modelfree.a2c import A2C
    from tianshou.algorithm.modelfree.reinforce import Reinforce

    out = {}
    variants = [
        dict(name="a2c_default", kind="a2c", kw={}, batch_size=None, repeat=1),
        dict(name="a2c_mb64", kind="a2c", kw=dict(vf_coef=0.25, ent_coef=0.02, max_grad_norm=0.5), batch_size=64, repeat=2),
        dict(name="reinforce_default", kind="reinforce", kw={}, batch_size=None, repeat=1),
        dict(name="reinforce_std_mb50", kind="reinforce", kw=dict(return_standardization=True), batch_size=50, repeat=1),
    ]
    n_env, T, obs_dim, n_act = 8, 25, 18, 5
    for v in variants:
        rng = np.random.default_rng(7)
        p = v["name"] + "_"
        torch.manual_seed(3)
        actor = DiscreteActor(preprocess_net=Net(state_shape=(obs_dim,), hidden_sizes=[64, 64]), action_shape=n_act,
                              softmax_output=False)
        critic = DiscreteCritic(preprocess_net=Net(state_shape=(obs_dim,), hidden_sizes=[64, 64]))
        for m in list(actor.modules()) + list(critic.modules()):
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.orthogonal_(m.weight)
                torch.nn.init.zeros_(m.bias)
        policy = DiscreteActorPolicy(actor=actor, action_space=gym.spaces.Discrete(n_act))
        if v["kind"] == "a2c":
            algo = A2C(policy=policy, critic=critic, optim=AdamOptimizerFactory(lr=3e-4), **v["kw"])
        else:
            algo = Reinforce(policy=policy, optim=AdamOptimizerFactory(lr=3e-4), **v["kw"])
        buf = fill_vector_buffer(rng, n_env, T, obs_dim, p_term=0.03, trunc_at=25, n_act=n_act)
        batch, indices = buf.sample(0)
        out.update({p + k: val for k, val in net_params(actor, critic).items()})
        out.update({p + "obs": batch.obs, p + "obs_next": batch.obs_next, p + "act": batch.act, p + "rew": batch.rew,
                    p + "terminated": batch.terminated, p + "truncated": batch.truncated, p + "indices": indices,
                    p + "unfinished": buf.unfinished_index()})
        if v["kw"].get("return_standardization"):
            rms = algo.discounted_return_computation.ret_rms
            rms.update(rng.standard_normal(50) * 3.0 + 1.5)  # non-trivial running stats (mean != 0 bootstraps cut episodes)
            out[p + "rms_before"] = np.array([rms.mean, rms.var, rms.count], np.float64)
        with policy_within_training_step(algo.policy):
            pb = algo._preprocess_batch(batch, buf, indices)
            out[p + "returns"] = np.asarray(pb.returns if isinstance(pb.returns, np.ndarray) else pb.returns.numpy()).copy()
            if v["kind"] == "a2c":
                out.update({p + "v_s": pb.v_s.numpy().copy(), p + "adv": pb.adv.numpy().copy()})
            if v["kw"].get("return_standardization"):
                out[p + "rms_after"] = np.array([rms.mean, rms.var, rms.count], np.float64)
            np.random.seed(11)
            st = np.random.get_state()
            out[p + "perms"] = np.stack([np.random.permutation(len(indices)) for _ in range(v["repeat"])])
            np.random.set_state(st)
            with torch.enable_grad():
                algo.train()
                stats = algo._update_with_batch(pb, v["batch_size"], v["repeat"])
        out[p + "batch_size"] = -1 if v["batch_size"] is None else v["batch_size"]
        out[p + "repeat"] = v["repeat"]
        for k in ("loss", "actor_loss", "vf_loss", "ent_loss"):
            if hasattr(stats, k):
                s = getattr(stats, k)
                out[p + "stat_" + k] = np.array([s.mean, s.std, s.max, s.min], np.float64)
        out.update({p + "after_" + k: val for k, val in net_params(actor, critic).items()})
        lin = lambda m: [x for x in m.modules() if isinstance(x, torch.nn.Linear)]  # noqa: E731
        for i, l in enumerate(lin(actor)):  # gradients of the last gradient step (the critic has none under Reinforce)
            out[p + f"last_actor_gw{i}"] = l.weight.grad.detach().numpy().copy()
            out[p + f"last_actor_gb{i}"] = l.bias.grad.detach().numpy().copy()
        out[p + "cfg"] = np.array([getattr(algo, "vf_coef", 0.0), getattr(algo, "ent_coef", 0.0),
                                   v["kw"].get("max_grad_norm") or 0.0, 3e-4, 0.99,
                                   getattr(algo, "gae_lambda", 1.0)], np.float64)
    save("pg_update.npz", variants=np.array([v["name"] for v in variants]), **out)


# ------------------------------------------------------------------------------------------------
class _FakeAECEnv:
    """Minimal stand-in exposing what MARLDispatcher reads (marl.py:197-203)."""

    def __init__(self, n):
        self.agents = [f"agent_{i}" for i in range(n)]
        self.agent_idx = {a: i for i, a in enumerate(self.agents)}


class _ArgmaxPolicy(Policy):
    """Deterministic mock policy: act = argmax(obs @ W) (lets the scatter be pinned exactly)."""

    def __init__(self, W, n_act):
        super().__init__(action_space=gym.spaces.Discrete(n_act))
        self.W = torch.nn.Parameter(torch.from_numpy(W), requires_grad=False)
        self.calls = 0

    def forward(self, batch, state=None, **kw):
        self.calls += 1
        obs = batch.obs.obs if hasattr(batch.obs, "obs") else batch.obs
        logits = torch.as_tensor(obs, dtype=torch.float32) @ self.W
        return Batch(act=logits.argmax(-1).numpy(), state=None, logits=logits.numpy())


def make_marl_dispatch() -> None:
    rng = np.random.default_rng(5)
    out = {}
    N, obs_dim, n_act, B = 3, 6, 5, 37
    env = _FakeAECEnv(N)
    Ws = [rng.standard_normal((obs_dim, n_act)).astype(np.float32) for _ in range(N)]
    agent_rows = rng.integers(0, N, B)
    agent_id = np.array([env.agents[i] for i in agent_rows], dtype=object)
    obs = rng.standard_normal((B, obs_dim)).astype(np.float32)
    batch = Batch(obs=Batch(agent_id=agent_id, obs=obs), info=Batch())
    # independent policies via MultiAgentPolicy.forward (marl.py:108-185)
    pols = {a: _ArgmaxPolicy(Ws[i], n_act) for i, a in enumerate(env.agents)}
    res = MultiAgentPolicy(pols)(batch)
    out.update(agent_rows=agent_rows, obs=obs, Ws=np.stack(Ws), act_independent=np.asarray(res.act))
    # shared mode via FlexibleMultiAgentPolicyManager (flexible_policy.py:189-231): ONE forward
    shared = _ArgmaxPolicy(Ws[0], n_act)
    mgr = FlexibleMultiAgentPolicyManager(policies=shared, env=env, mode="shared")
    res = mgr(batch)
    out.update(act_shared=np.asarray(res.act), shared_calls=shared.calls)
    # grouped mode: agents 0,1 -> policy A ; agent 2 -> policy B
    pa, pb = _ArgmaxPolicy(Ws[1], n_act), _ArgmaxPolicy(Ws[2], n_act)
    mgr = FlexibleMultiAgentPolicyManager(
        policies={"g0": pa, "g1": pb}, env=env, mode="grouped",
        agent_groups={"g0": ["agent_0", "agent_1"], "g1": ["agent_2"]})
    res = mgr(batch)
    # Reference defect (quirk Q6): grouped mode re-keys `self.policies` by GROUP name
    # (flexible_policy.py:96-98) and then falls back to MultiAgentPolicy.forward (:200), which
    # matches `obs.agent_id == <group name>` (marl.py:148) -> no rows match, no `act` is produced.
    out.update(grouped_forward_has_act=int("act" in res.get_keys()),
               grouped_policy_of_agent=np.array([0, 0, 1]))
    # the intended semantics (policy_map, flexible_policy.py:138-158) evaluated by hand:
    exp = np.zeros(B, np.int64)
    for a_i, pol in enumerate([pa, pa, pb]):
        rows = np.nonzero(agent_rows == a_i)[0]
        exp[rows] = pol(Batch(obs=obs[rows])).act
    out.update(act_grouped_policy_map=exp)

    # ---- MARLDispatcher per-agent GAE over AEC rows incl. quirk Q1 ----
    n_env, T = 2, 3
    buf = VectorReplayBuffer(n_env * T * 2, n_env)
    k = 0
    for t in range(T):
        for a in range(2):  # AEC: agents alternate turns, one row per turn
            k += 1
            rew = np.zeros((n_env, 2))
            rew[:, a] = np.arange(1, n_env + 1) * k
            ids = np.array([f"agent_{a}"] * n_env, dtype=object)
            b = Batch(obs=Batch(agent_id=ids, obs=np.zeros((n_env, 2), np.float32)),
                      act=np.zeros(n_env, int), rew=rew,
                      terminated=np.zeros(n_env, bool), truncated=np.zeros(n_env, bool),
                      obs_next=Batch(agent_id=ids, obs=np.zeros((n_env, 2), np.float32)))
            buf.add(b, buffer_ids=np.arange(n_env))
    batch, indices = buf.sample(0)
    env2 = _FakeAECEnv(2)
    q1 = {}
    for a_i, agent in enumerate(env2.agents):
        idx = np.nonzero(batch.obs.agent_id == agent)[0]
        tmp, tind = batch[idx], indices[idx]
        tmp.rew = tmp.rew[:, a_i]
        save_rew, buf._meta.rew = buf.rew, Batch()
        buf._meta.rew = save_rew[:, a_i]
        ret, adv = Algorithm.compute_episodic_return(tmp, buf, tind, np.zeros(len(idx)), np.zeros(len(idx)),
                                                     gamma=1.0, gae_lambda=1.0)
        buf._meta.rew = save_rew
        q1[f"q1_agent{a_i}_idx"] = idx
        q1[f"q1_agent{a_i}_indices"] = tind
        q1[f"q1_agent{a_i}_rew"] = np.asarray(tmp.rew)
        q1[f"q1_agent{a_i}_returns"] = ret
    out.update(q1, q1_unfinished=buf.unfinished_index(), q1_all_indices=indices,
               q1_agent_of_row=np.array([0 if a == "agent_0" else 1 for a in batch.obs.agent_id]))
    save("marl_dispatch.npz", **out)


# ------------------------------------------------------------------------------------------------
def make_ctde() -> None:
    rng = np.random.default_rng(9)
    torch.manual_seed(9)
    N, obs_dim, n_act, B, hid = 3, 6, 5, 32, 16
    out = {}
    obs_by_agent = {f"agent_{i}": torch.from_numpy(rng.standard_normal((B, obs_dim)).astype(np.float32))
                    for i in range(N)}
    for mode in ("concatenate", "mean"):
        g = GlobalStateConstructor(mode=mode, obs_dim=obs_dim, n_agents=N).build(obs_by_agent)
        out["global_" + mode] = g.numpy()
    out["obs_by_agent"] = np.stack([v.numpy() for v in obs_by_agent.values()])
    actor = DecentralizedActor(obs_dim, n_act, hidden_dim=hid)
    critic = CentralizedCritic(N * obs_dim, N, hidden_dim=hid)
    pol = CTDEPolicy(actor=actor, critic=critic,
                     optim_actor=torch.optim.Adam(actor.parameters(), lr=1e-3),
                     optim_critic=torch.optim.Adam(critic.parameters(), lr=1e-3),
                     observation_space=gym.spaces.Box(-np.inf, np.inf, (obs_dim,)),
                     action_space=gym.spaces.Discrete(n_act))
    for name, mod in (("actor", actor), ("critic", critic)):
        for i, l in enumerate([mod.fc1, mod.fc2, mod.fc3]):
            out[f"{name}_w{i}"] = l.weight.detach().numpy().copy()
            out[f"{name}_b{i}"] = l.bias.detach().numpy().copy()
    batch = Batch(
        obs=obs_by_agent["agent_0"].numpy(), act=rng.integers(0, n_act, B),
        rew=rng.standard_normal(B).astype(np.float32),
        obs_next=rng.standard_normal((B, obs_dim)).astype(np.float32),
        terminated=rng.random(B) < 0.1,
        global_obs=out["global_concatenate"],
        global_obs_next=rng.standard_normal((B, N * obs_dim)).astype(np.float32))
    fwd = pol.forward(Batch(obs=batch.obs))
    out["fwd_act_logits"] = fwd.act.detach().numpy().copy()
    losses = pol.learn(batch)
    out.update(b_obs=batch.obs, b_act=batch.act, b_rew=batch.rew, b_obs_next=batch.obs_next,
               b_terminated=batch.terminated, b_global_obs=batch.global_obs,
               b_global_obs_next=batch.global_obs_next,
               actor_loss=losses["actor_loss"], critic_loss=losses["critic_loss"])
    for name, mod in (("actor", actor), ("critic", critic)):
        for i, l in enumerate([mod.fc1, mod.fc2, mod.fc3]):
            out[f"after_{name}_w{i}"] = l.weight.detach().numpy().copy()
            out[f"after_{name}_b{i}"] = l.bias.detach().numpy().copy()
            out[f"grad_{name}_w{i}"] = l.weight.grad.detach().numpy().copy()
            out[f"grad_{name}_b{i}"] = l.bias.grad.detach().numpy().copy()
    save("ctde.npz", **out)


def _make_ctde_rows(fname: str, N: int, D: int, A: int, H: int, E: int, T: int, first_seed: int) -> None:
    """CTDEPolicy.learn (ctde.py:121-199) with 128-wide nets and the critic on the N*D joint row, called the way the MARL
    trainers call it (training_coordinator.py:118: one learn() per agent, in env.agents order, on that agent's column of the
    same joint rows).  The rows are generated as a time-major store [T, E, N, ...] whose obs_next is the next slot's obs unless
    the episode ended (what a Collector leaves behind, collector.py:1040-1069); the reference gets the env-major flattening.
    Variant `chain`: episodes end at the last slot only; `early`: some end mid-store."""
    out = dict(dims=np.array([N, D, A, H, E, T]), gamma=np.float64(0.99), lr=np.float64(1e-3))

    def run(variant, seed):
        res = {}
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        obs = rng.standard_normal((T, E, N, D)).astype(np.float32)
        act = rng.integers(0, A, (T, E, N))
        rew = rng.standard_normal((T, E, N)).astype(np.float32)
        done = np.zeros((T, E), bool)
        done[T - 1] = rng.random(E) < 0.6
        if variant == "early":
            done[3, 1] = done[6, 4] = done[7, 1] = done[0, 5] = True
        is_term = rng.random((T, E)) < 0.5  # an ended episode is terminated for every agent or truncated for every agent
        term = np.repeat((done & is_term)[:, :, None], N, axis=2)
        trunc = np.repeat((done & ~is_term)[:, :, None], N, axis=2)
        obs_next = np.empty_like(obs)
        obs_next[:-1] = obs[1:]
        fresh = rng.standard_normal((T, E, N, D)).astype(np.float32)  # the final observation of an episode / of the store
        obs_next[T - 1] = fresh[T - 1]
        obs_next[done] = fresh[done]
        actor = DecentralizedActor(D, A, hidden_dim=H)
        critic = CentralizedCritic(N * D, N, hidden_dim=H)
        pol = CTDEPolicy(actor=actor, critic=critic,
                         optim_actor=torch.optim.Adam(actor.parameters(), lr=1e-3),
                         optim_critic=torch.optim.Adam(critic.parameters(), lr=1e-3),
                         observation_space=gym.spaces.Box(-np.inf, np.inf, (D,)),
                         action_space=gym.spaces.Discrete(A), discount_factor=0.99)

        def snap(tag):
            for name, mod in (("actor", actor), ("critic", critic)):
                for i, l in enumerate([mod.fc1, mod.fc2, mod.fc3]):
                    res[f"{variant}_{tag}_{name}_w{i}"] = l.weight.detach().numpy().copy()
                    res[f"{variant}_{tag}_{name}_b{i}"] = l.bias.detach().numpy().copy()

        def min_preact(mod, x):  # how close any hidden unit of the reference's own run comes to the ReLU kink
            with torch.no_grad():
                z1 = mod.fc1(torch.as_tensor(x))
                z2 = mod.fc2(torch.relu(z1))
            return min(float(z1.abs().min()), float(z2.abs().min()))

        snap("init")
        em = lambda x: np.ascontiguousarray(np.swapaxes(x, 0, 1))  # noqa: E731  [E, T, ...]: env-major rows
        losses, minz = [], np.inf
        for a in range(N):
            batch = Batch(obs=em(obs)[:, :, a].reshape(E * T, D), act=em(act)[:, :, a].reshape(E * T),
                          rew=em(rew)[:, :, a].reshape(E * T), obs_next=em(obs_next)[:, :, a].reshape(E * T, D),
                          terminated=em(term)[:, :, a].reshape(E * T),
                          global_obs=em(obs).reshape(E * T, N * D), global_obs_next=em(obs_next).reshape(E * T, N * D))
            minz = min(minz, min_preact(actor, batch.obs), min_preact(critic, batch.global_obs), min_preact(critic, batch.global_obs_next))
            r = pol.learn(batch)
            losses.append([r["actor_loss"], r["critic_loss"]])
            # How far one Adam step moves a parameter per unit of gradient error, summed over the calls, from the reference's
            # own optimizer state: u_k = lr m^_k / (sqrt(v^_k) + eps), |du_k / dg| <= lr / (sqrt(v^_k) + eps).  A parameter
            # whose gradients all but cancel (|g| ~ eps) takes a sign-like step of up to lr whatever the summation order says;
            # the replay tests allow `adamcond` x (their gradient bar) on top of the plain weight tolerance.
            for name, mod, opt in (("actor", actor, pol.optim_actor), ("critic", critic, pol.optim_critic)):
                for i, l in enumerate([mod.fc1, mod.fc2, mod.fc3]):
                    for kind, prm in (("w", l.weight), ("b", l.bias)):
                        st = opt.state[prm]
                        v_hat = st["exp_avg_sq"].detach().double().numpy() / (1.0 - 0.999 ** float(st["step"]))
                        key = f"{variant}_adamcond_{name}_{kind}{i}"
                        res[key] = res.get(key, 0.0) + 1e-3 / (np.sqrt(v_hat) + 1e-8)
            if a == 0:
                snap("after1")
                for name, mod in (("actor", actor), ("critic", critic)):
                    for i, l in enumerate([mod.fc1, mod.fc2, mod.fc3]):
                        res[f"{variant}_grad1_{name}_w{i}"] = l.weight.grad.detach().numpy().copy()
                        res[f"{variant}_grad1_{name}_b{i}"] = l.bias.grad.detach().numpy().copy()
        snap("afterN")
        for k in [k for k in res if "_adamcond_" in k]:
            res[k] = res[k].astype(np.float32)
        res.update({f"{variant}_obs": obs, f"{variant}_act": act, f"{variant}_rew": rew, f"{variant}_term": term,
                    f"{variant}_trunc": trunc, f"{variant}_obs_next": obs_next, f"{variant}_losses": np.array(losses),
                    f"{variant}_seed": np.int64(seed), f"{variant}_min_abs_preact": np.float64(minz)})
        return res, minz

    # A hidden unit whose pre-activation is ~1e-7 on some row takes either side of the ReLU depending on the summation order of
    # a K1-term f32 dot product; Adam then turns that one row's gradient into full +-lr steps on the unit's whole weight row
    # (seed 21 of `chain` has |z| = 1.06e-7 in the fourth call: f64 and the reference's f32 land on one side, a k-ordered f32 FMA
    # chain on the other).  The fixture is for arithmetic parity, not for tie-breaking at a kink: take the first seed whose
    # every pre-activation (reference's own f32 run, all N calls, both nets) stays 2e-6 away from it; the margin is recorded.
    # (The criterion is computed from the reference's run alone; the fixture is therefore silent about tie-breaking at a kink.)
    for variant in ("chain", "early"):
        for seed in range(first_seed, first_seed + 400):
            res, minz = run(variant, seed)
            if minz >= 2e-6:
                break
        else:
            raise RuntimeError(f"{fname} {variant}: no seed keeps every pre-activation 2e-6 off the kink")
        print(f"{fname} {variant}: seed {seed}, min |pre-activation| {minz:.2e}")
        out.update(res)
    save(fname, **out)


def make_ctde_wide() -> None:
    """128-wide nets at N = 4, D = 24: critic input K1 = 96 (the <6> instantiations of the critic kernels), n_out = 4."""
    _make_ctde_rows("ctde_wide.npz", N=4, D=24, A=5, H=128, E=6, T=11, first_seed=21)


def make_ctde_c3() -> None:
    """BASELINE configs[2]'s own widths: N = 8 agents, obs 48 -> critic input K1 = 384, n_out = 8 (ctde.py:346-414) -- the
    <24> instantiations of critic_rows_train_kernel / critic_dw1_kernel / critic_rows_forward_kernel, the ones the bench runs."""
    _make_ctde_rows("ctde_c3.npz", N=8, D=48, A=5, H=128, E=6, T=11, first_seed=21)


# ------------------------------------------------------------------------------------------------
sys.path.insert(0, HERE)
from async_script import scripted_ready  # noqa: E402  (shared with the replay test)


def make_async_collector() -> None:
    """AsyncCollector (collector.py:1116-1394) over an async vector env whose readiness is SCRIPTED (`scripted_ready`, through the
    worker class's `wait`, venvs.py:301): four MoveToRight envs of lengths 2..5, wait_num 3, a policy that always moves right.
    A sequence of collect(n_episode=...) / collect(n_step=...) calls; after each: the statistics and the ready set; at the end
    the whole buffer."""
    from tianshou.data import AsyncCollector
    from tianshou.env import BaseVectorEnv
    from tianshou.env.worker import DummyEnvWorker

    class MoveToRight(gym.Env):
        def __init__(self, size):
            self.size, self.index = size, 0
            self.action_space = gym.spaces.Discrete(2)
            self.observation_space = gym.spaces.Box(0, size, (1,))

        def reset(self, seed=None, options=None):
            self.index = 0
            return np.array([self.index], np.float32), {"key": 1}

        def step(self, action):
            self.index = self.index + 1 if int(action) == 1 else max(0, self.index - 1)
            done = self.index == self.size
            return np.array([self.index], np.float32), float(done) * (self.size + 1), done, False, {"key": 1}

    calls = [0]

    class ScriptedWorker(DummyEnvWorker):
        @staticmethod
        def wait(workers, wait_num, timeout=None):
            pos = scripted_ready(len(workers), wait_num, calls[0])
            calls[0] += 1
            return [workers[p_] for p_ in pos]

    class RightPolicy(Policy):
        """Always moves right; its `policy` entry tags every action with (forward call number, position in the call): the stored
        rows then show that act / policy entries are the ones handed out for THAT env, whichever call's step returned it."""

        def __init__(self):
            super().__init__(action_space=gym.spaces.Discrete(2))
            self.calls = 0

        def forward(self, batch, state=None, **kw):
            self.calls += 1
            n = len(batch.obs)
            # (a hidden state is returned because the reference's AsyncCollector cannot run without one: collector.py:1337 indexes
            # `_current_hidden_state_in_all_envs_EH` unconditionally -- quirk Q8; the build's AsyncCollector has no such need)
            return Batch(act=np.ones(n, np.int64), state=np.zeros((n, 1), np.float32),
                         policy=Batch(logp=(self.calls * 10.0 + np.arange(n)).astype(np.float32)))

    sizes = [2, 3, 4, 5]
    venv = BaseVectorEnv([lambda s=s: MoveToRight(s) for s in sizes], ScriptedWorker, wait_num=3)
    assert venv.is_async
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        col = AsyncCollector(RightPolicy(), venv, VectorReplayBuffer(total_size=240, buffer_num=4))
    col.reset()
    trace = []  # env ids handed to / returned by every venv.step call: the interleaving itself
    orig_step = venv.step

    def logged_step(action, id=None):  # noqa: A002
        out_ = orig_step(action, id)
        trace.append((list(np.asarray(id).tolist()) if id is not None else [], [int(i["env_id"]) for i in out_[-1]]))
        return out_

    venv.step = logged_step
    plan = [("n_episode", 1), ("n_episode", 3), ("n_step", 1), ("n_step", 5), ("n_episode", 2), ("n_step", 7), ("n_step", 10),
            ("n_episode", 4)]
    out = dict(sizes=np.array(sizes), wait_num=np.int64(3), plan_kind=np.array([k for k, _ in plan]),
               plan_n=np.array([n for _, n in plan]))
    for i, (kind, n) in enumerate(plan):
        st = col.collect(**{kind: n})
        out[f"c{i}_steps"], out[f"c{i}_episodes"] = np.int64(st.n_collected_steps), np.int64(st.n_collected_episodes)
        out[f"c{i}_lens"], out[f"c{i}_returns"] = np.asarray(st.lens, np.int64), np.asarray(st.returns, np.float64)
        out[f"c{i}_ready"] = np.asarray(col._ready_env_ids_R, np.int64)
        out[f"c{i}_len_buf"] = np.int64(len(col.buffer))
        out[f"c{i}_waiting"] = np.asarray(sorted(venv.waiting_id), np.int64)
    buf = col.buffer
    idx = buf.sample_indices(0)
    b = buf[idx]
    out.update(indices=idx, obs=np.asarray(b.obs), obs_next=np.asarray(b.obs_next), act=np.asarray(b.act), rew=np.asarray(b.rew),
               terminated=np.asarray(b.terminated), truncated=np.asarray(b.truncated), done=np.asarray(b.done),
               env_id=np.asarray(b.info.env_id), policy_tag=np.asarray(b.policy.logp), wait_calls=np.int64(calls[0]),
               trace_sent=np.array([x for snt, _ in trace for x in snt + [-1]], np.int64),
               trace_returned=np.array([x for _, ret in trace for x in ret + [-1]], np.int64),
               collect_step=np.int64(col.collect_step), collect_episode=np.int64(col.collect_episode))
    save("async_collector.npz", **out)


def make_collector() -> None:
    """The synchronous Collector (collector.py:770-1098) through a scripted sequence of calls (collector_script.PLAN): five
    MoveToRight envs, two of them with a step limit (truncation), a policy whose actions follow `scripted_action` and whose
    `policy` entry tags every action with (forward call, position).  n_step calls (a multiple of the env count and not),
    n_episode calls with fewer and with more episodes than envs (the surplus-env removal of Step 13, the env reset after an
    n_episode call), reset_before_collect, reset_buffer, reset_stat.  After each call: the statistics, the counters, the
    observations the next call starts from; at the end every buffer row."""
    from collector_script import LIMITS, PLAN, SIZES, env_step, scripted_action
    from tianshou.data import Collector
    from tianshou.env import DummyVectorEnv

    class MoveToRight(gym.Env):
        def __init__(self, size, limit):
            self.size, self.limit, self.index, self.steps = size, limit, 0, 0
            self.action_space = gym.spaces.Discrete(2)
            self.observation_space = gym.spaces.Box(0, size, (1,))

        def reset(self, seed=None, options=None):
            self.index, self.steps = 0, 0
            return np.array([self.index], np.float32), {"key": 1}

        def step(self, action):
            self.index, self.steps, rew, term, trunc = env_step(self.index, self.steps, self.size, self.limit, action)
            return np.array([self.index], np.float32), rew, term, trunc, {"key": 1}

    class ScriptPolicy(Policy):
        def __init__(self):
            super().__init__(action_space=gym.spaces.Discrete(2))
            self.calls = 0

        def forward(self, batch, state=None, **kw):
            self.calls += 1
            obs = np.asarray(batch.obs)
            return Batch(act=scripted_action(self.calls, obs),
                         policy=Batch(logp=(self.calls * 10.0 + np.arange(len(obs))).astype(np.float32)))

    venv = DummyVectorEnv([lambda s=s, l=l: MoveToRight(s, l) for s, l in zip(SIZES, LIMITS)])
    pol = ScriptPolicy()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        col = Collector(pol, venv, VectorReplayBuffer(total_size=400, buffer_num=len(SIZES)))
        col.reset()
        out = dict(sizes=np.array(SIZES), limits=np.array(LIMITS))
        for i, (kind, n, extra) in enumerate(PLAN):
            kw = {}
            if extra == "reset_before_collect":
                kw["reset_before_collect"] = True
            elif extra == "reset_buffer":
                col.reset_buffer()
            elif extra == "reset_stat":
                col.reset_stat()
            st = col.collect(**{kind: n}, **kw)
            out[f"c{i}_steps"], out[f"c{i}_episodes"] = np.int64(st.n_collected_steps), np.int64(st.n_collected_episodes)
            out[f"c{i}_lens"], out[f"c{i}_returns"] = np.asarray(st.lens, np.int64), np.asarray(st.returns, np.float64)
            out[f"c{i}_len_buf"] = np.int64(len(col.buffer))
            out[f"c{i}_counters"] = np.array([col.collect_step, col.collect_episode, pol.calls], np.int64)
            out[f"c{i}_pre_obs"] = np.asarray(col._pre_collect_obs_RO, np.float32).reshape(-1)
            if st.returns_stat is not None:
                out[f"c{i}_ret_stat"] = np.array([st.returns_stat.mean, st.returns_stat.std, st.returns_stat.max, st.returns_stat.min])
                out[f"c{i}_len_stat"] = np.array([st.lens_stat.mean, st.lens_stat.std, st.lens_stat.max, st.lens_stat.min])
    buf = col.buffer
    idx = buf.sample_indices(0)
    b = buf[idx]
    out.update(indices=idx, obs=np.asarray(b.obs), obs_next=np.asarray(b.obs_next), act=np.asarray(b.act), rew=np.asarray(b.rew),
               terminated=np.asarray(b.terminated), truncated=np.asarray(b.truncated), done=np.asarray(b.done),
               policy_tag=np.asarray(b.policy.logp), last_index=np.asarray(buf.last_index, np.int64))
    save("collector.npz", **out)


def make_collector_port() -> None:
    """The synchronous Collector + VectorReplayBuffer in the ONE mode bench.py's CPU baseline port restates
    (oracle/cpu_path.py::PortCollector: collect(n_step) with every env ready, reset_buffer(keep_statistics=True) between calls as
    the trainer does, trainer.py:1104): the truncating MoveToRight envs and the scripted policy of collector_script.py, three
    n_step calls.  After each call: the statistics and every buffer row in sample_indices(0) order."""
    from collector_script import LIMITS, SIZES, env_step, scripted_action
    from tianshou.data import Collector
    from tianshou.env import DummyVectorEnv

    class MoveToRight(gym.Env):
        def __init__(self, size, limit):
            self.size, self.limit, self.index, self.steps = size, limit, 0, 0
            self.action_space = gym.spaces.Discrete(2)
            self.observation_space = gym.spaces.Box(0, size, (1,))

        def reset(self, seed=None, options=None):
            self.index, self.steps = 0, 0
            return np.array([self.index], np.float32), {}

        def step(self, action):
            self.index, self.steps, rew, term, trunc = env_step(self.index, self.steps, self.size, self.limit, action)
            return np.array([self.index], np.float32), rew, term, trunc, {}

    class ScriptPolicy(Policy):
        def __init__(self):
            super().__init__(action_space=gym.spaces.Discrete(2))
            self.calls = 0

        def forward(self, batch, state=None, **kw):
            self.calls += 1
            return Batch(act=scripted_action(self.calls, np.asarray(batch.obs)))

    venv = DummyVectorEnv([lambda s=s, l=l: MoveToRight(s, l) for s, l in zip(SIZES, LIMITS)])
    pol = ScriptPolicy()
    import warnings

    out = dict(sizes=np.array(SIZES), limits=np.array(LIMITS), n_steps=np.array([10, 15, 20]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        col = Collector(pol, venv, VectorReplayBuffer(total_size=len(SIZES) * 8, buffer_num=len(SIZES)))
        col.reset()
        for i, n in enumerate(out["n_steps"]):
            if i:
                col.reset_buffer(keep_statistics=True)
            st = col.collect(n_step=int(n))
            buf = col.buffer
            idx = buf.sample_indices(0)
            b = buf[idx]
            out.update({f"c{i}_steps": np.int64(st.n_collected_steps), f"c{i}_episodes": np.int64(st.n_collected_episodes),
                        f"c{i}_lens": np.asarray(st.lens, np.int64), f"c{i}_returns": np.asarray(st.returns, np.float64),
                        f"c{i}_counters": np.array([col.collect_step, col.collect_episode, pol.calls], np.int64),
                        f"c{i}_indices": idx, f"c{i}_obs": np.asarray(b.obs), f"c{i}_obs_next": np.asarray(b.obs_next),
                        f"c{i}_act": np.asarray(b.act), f"c{i}_rew": np.asarray(b.rew),
                        f"c{i}_terminated": np.asarray(b.terminated), f"c{i}_truncated": np.asarray(b.truncated)})
    save("collector_port.npz", **out)


def make_trainers() -> None:
    """The training coordinators (training_coordinator.py:27-760) as a TRACE: mock policies that count `learn` calls, seeded numpy
    randomness.  SimultaneousTrainer with per-agent frequencies, SequentialTrainer with a custom order, SelfPlayTrainer (snapshot
    pool, the three opponent-sampling rules in turn, win-rate updates) and LeaguePlayTrainer (the three matchmaking rules in turn,
    Elo / performance updates, promotion / relegation lists) -- who learns at every step, which snapshot is sampled, every rating.
    The replay must draw from numpy's global generator call for call as the reference does."""
    import trainer_script as ts
    from tianshou.algorithm.multiagent.flexible_policy import FlexibleMultiAgentPolicyManager
    from tianshou.algorithm.multiagent.training_coordinator import (LeaguePlayTrainer, SelfPlayTrainer, SequentialTrainer,
                                                                    SimultaneousTrainer)

    class MockPolicy(Policy):
        def __init__(self):
            super().__init__(observation_space=gym.spaces.Box(-1, 1, (4,)), action_space=gym.spaces.Discrete(2))
            self.learn_count, self.version = 0, -1

        def forward(self, batch, state=None, **kw):
            return Batch(act=np.zeros(len(batch.obs), np.int64), state=state)

        def learn(self, batch, **kw):
            self.learn_count += 1
            return {"loss": float(self.learn_count)}

    class Env:
        def __init__(self, n):
            self.agents = [f"agent_{i}" for i in range(n)]
            self.observation_space, self.action_space = gym.spaces.Box(-1, 1, (4,)), gym.spaces.Discrete(2)

    def ma_batch(env):
        return Batch({a: Batch(obs=np.zeros((4, 4), np.float32), act=np.zeros(4, np.int64), rew=np.ones(4),
                               terminated=np.zeros(4, bool), truncated=np.zeros(4, bool), obs_next=np.zeros((4, 4), np.float32),
                               info=Batch()) for a in env.agents})

    out = {}
    # -- simultaneous / sequential: who learned at each step
    env = Env(3)
    pols = {a: MockPolicy() for a in env.agents}
    tr = SimultaneousTrainer(FlexibleMultiAgentPolicyManager(pols, env, mode="independent"), agent_train_freq={"agent_1": 2, "agent_2": 3})
    rows = []
    for _ in range(ts.SIMULTANEOUS_STEPS):
        losses = tr.train_step(ma_batch(env))
        rows.append([int(a in losses) for a in env.agents] + [pols[a].learn_count for a in env.agents])
    out["simultaneous"] = np.array(rows, np.int64)
    pols = {a: MockPolicy() for a in env.agents}
    tr = SequentialTrainer(FlexibleMultiAgentPolicyManager(pols, env, mode="independent"), agent_order=["agent_2", "agent_0", "agent_1"],
                           steps_per_agent=2)
    rows = []
    for _ in range(ts.SEQUENTIAL_STEPS):
        losses = tr.train_step(ma_batch(env))
        rows.append([env.agents.index(next(iter(losses)))] + [int(pols[a].training) for a in env.agents])
    out["sequential"] = np.array(rows, np.int64)
    # -- self-play
    env = Env(2)
    pols = {a: MockPolicy() for a in env.agents}
    tr = SelfPlayTrainer(FlexibleMultiAgentPolicyManager(pols, env, mode="independent"), main_agent_id="agent_0",
                         snapshot_interval=ts.SNAPSHOT_INTERVAL, opponent_pool_size=ts.POOL_SIZE)
    np.random.seed(ts.SEED)
    rows, rates = [], []
    for i in range(ts.SELFPLAY_STEPS):
        pols["agent_0"].version = i
        tr.opponent_sampling = ts.selfplay_sampling(i)
        losses = tr.train_step(ma_batch(env))
        opp = tr._sample_opponent()
        if opp is not None:
            tr.update_win_rate(id(opp), ts.selfplay_won(i))
        pool = [p.version for p in tr.opponent_pool] + [-1] * (ts.POOL_SIZE - len(tr.opponent_pool))
        rows.append([int("agent_0" in losses), pols["agent_0"].learn_count, pols["agent_1"].learn_count,
                     -1 if opp is None else opp.version, len(tr.opponent_win_rates), *pool])
        rates.append([tr.opponent_win_rates.get(id(p), -1.0) for p in tr.opponent_pool] + [-1.0] * (ts.POOL_SIZE - len(tr.opponent_pool)))
    out["selfplay"], out["selfplay_rates"] = np.array(rows, np.int64), np.array(rates, np.float64)
    # -- league
    env = Env(ts.LEAGUE_AGENTS)
    pols = {a: MockPolicy() for a in env.agents}
    tr = LeaguePlayTrainer(FlexibleMultiAgentPolicyManager(pols, env, mode="independent"), games_per_evaluation=ts.GAMES_PER_EVALUATION)
    np.random.seed(ts.SEED)
    rows, elo, perf, lists = [], [], [], []
    for i in range(ts.LEAGUE_STEPS):
        tr.matchmaking = ts.league_matchmaking(i)
        before = [pols[a].learn_count for a in env.agents]
        losses = tr.train_step(ma_batch(env))
        match = [a for a in env.agents if pols[a].learn_count > before[env.agents.index(a)]]
        order = list(losses)  # (the match, in the order the trainer walked it)
        w, l = (order[0], order[1]) if ts.league_winner_first(i) else (order[1], order[0])
        tr.update_match_result(w, l)
        rows.append([env.agents.index(order[0]), env.agents.index(order[1]), len(match), tr.game_count, len(tr.match_history)])
        elo.append([tr.elo_ratings[a] for a in env.agents])
        perf.append([tr.agent_performance[a] for a in env.agents])
        pr, rl = tr._update_league()
        lists.append([int(a in pr) for a in env.agents] + [int(a in rl) for a in env.agents])
    out.update(league=np.array(rows, np.int64), league_elo=np.array(elo, np.float64), league_perf=np.array(perf, np.float64),
               league_lists=np.array(lists, np.int64))
    save("trainers.npz", **out)


# ------------------------------------------------------------------------------------------------
def make_misc() -> None:
    out = {}
    rows = []
    for length in (1, 5, 63, 64, 65, 127, 128, 129, 150, 192, 200, 4800):
        for size in (-1, 1, 7, 64, 100, 256, 5000):
            b = Batch(x=np.arange(length))
            sizes = [len(mb) for mb in b.split(size, shuffle=False, merge_last=True)]
            firsts = [int(mb.x[0]) for mb in b.split(size, shuffle=False, merge_last=True)]
            rows.append((length, size, len(sizes), sum(s * (i + 1) for i, s in enumerate(sizes)),
                         sum(f * (i + 1) for i, f in enumerate(firsts))))
    out["split_rows"] = np.array(rows, np.int64)
    rng = np.random.default_rng(2)
    rms = RunningMeanStd()
    xs, states = [], []
    for n in (10, 1, 33, 100):
        x = rng.standard_normal(n) * 2.5 + 1.0
        rms.update(x)
        xs.append(x)
        states.append([float(rms.mean), float(rms.var), float(rms.count)])
    out["rms_x"] = np.concatenate(xs)
    out["rms_lens"] = np.array([len(x) for x in xs])
    out["rms_states"] = np.array(states)
    r = rng.standard_normal(17)
    out["mc_rew"] = r
    out["mc_ret"] = episode_mc_return_to_go(r, 0.97)
    save("misc.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gae", "vrb_trace", "ppo_update", "ppo_update_wide", "pg_update", "marl_dispatch", "ctde", "ctde_wide",
                             "ctde_c3", "async_collector", "collector", "collector_port", "trainers", "misc"]
    for w in which:
        globals()["make_" + w]()
