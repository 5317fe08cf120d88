"""TEST / BENCHMARK INFRASTRUCTURE ONLY -- CPU port of one bench.py step, in the reference's loop structure.

Used by bench.py's `cpu_baseline` leg (kind = "port"): the reference itself is pure Python and never
travels to the GPU box, so the baseline is this restatement of its hot path, timed on the box's host
cores on a bounded sample of the same workload:

  rollout  -- per-env Python loop over env objects + np.stack of the results (DummyVectorEnv,
              /root/reference/tianshou/env/venvs.py:281-322), policy forward in PyTorch-CPU with
              Categorical.sample (modelfree/reinforce.py:167-192), device->host action copy
              (data/collector.py:736), buffer index bookkeeping (oracle.VectorReplayBufferIndex ==
              data/buffer/manager.py:131-193) + numpy fancy-index scatter (manager.py:180);
  update   -- critic passes in chunks of max_batchsize=256 (modelfree/a2c.py:121-127), single-thread
              serial GAE (oracle.gae_lanes == numba `_gae`, algorithm_base.py:1079-1134), logp_old,
              minibatch loop with autograd + clip-loss + Adam (modelfree/ppo.py:164-224), 4 .item() per
              minibatch (ppo.py:213-216), per-agent dispatch of the shared algorithm (multiagent/marl.py:251-268).
It is a reported baseline, not a target.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

import mpe_oracle
import oracle


def _mlp(d_in, h, d_out):
    net = torch.nn.Sequential(torch.nn.Linear(d_in, h), torch.nn.ReLU(), torch.nn.Linear(h, h), torch.nn.ReLU(),
                              torch.nn.Linear(h, d_out))
    for m in net:
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    return net


class PortStores:
    """The row stores of the port: one row per joint step, flat reference index = env * S + slot (manager.py:38-46)."""

    def __init__(self, n_env: int, slots: int, n_agent: int, obs_shape: tuple) -> None:
        rows = n_env * slots
        self.obs = np.zeros((rows, n_agent, *obs_shape), np.float32)
        self.obs_next = np.zeros_like(self.obs)
        self.act = np.zeros((rows, n_agent), np.int64)
        self.rew = np.zeros((rows, n_agent), np.float64)
        self.term = np.zeros((rows, n_agent), bool)
        self.trunc = np.zeros((rows, n_agent), bool)
        self.index = oracle.VectorReplayBufferIndex(rows, n_env, n_agent)


class PortCollector:
    """`Collector._collect` for `collect(n_step=...)` with every env ready (collector.py:854-1069), as `run_baseline` times it:
    envs are objects with `reset() -> obs [N, ...]` and `step(act [N]) -> (obs, rew [N], terminated [N], truncated [N])`, stepped
    one after the other and stacked (DummyVectorEnv, venvs.py:281-322).  Pinned to the REFERENCE's own Collector +
    VectorReplayBuffer by tests/test_oracle_golden.py::test_cpu_port_collect_leg_matches_the_reference_collector
    (tests/golden/collector_port.npz: rows, lengths, returns and counters of three n_step calls)."""

    def __init__(self, envs: list, stores: PortStores) -> None:
        self.envs, self.st = envs, stores
        self.last_obs = np.stack([e.reset() for e in envs]).astype(np.float32)
        self.collect_step = self.collect_episode = 0

    def collect(self, act_fn, n_step: int) -> dict:
        """Steps the envs until at least n_step transitions are stored (a multiple of the env count, collector.py:1046-1051);
        returns the statistics of this call in the reference's episode order (by vector step, then by env id)."""
        st, envs = self.st, self.envs
        n_env = len(envs)
        sub = st.obs.shape[0] // n_env
        steps, lens, rets = 0, [], []
        while steps < n_step:
            act = act_fn(self.last_obs)
            res = [envs[i].step(act[i]) for i in range(n_env)]            # venvs.py:281-287
            obs_next = np.stack([r[0] for r in res]).astype(np.float32)   # venvs.py:311-322
            rew = np.stack([r[1] for r in res])
            term = np.stack([r[2] for r in res])
            trunc = np.stack([r[3] for r in res])
            done = (term | trunc).any(1)
            ptr, ep_rew, _, ep_idx = st.index.add(rew, done)
            st.obs[ptr], st.obs_next[ptr], st.act[ptr] = self.last_obs, obs_next, act
            st.rew[ptr], st.term[ptr], st.trunc[ptr] = rew, term, trunc
            steps += n_env
            self.last_obs = obs_next.copy()
            for i in np.where(done)[0]:
                # CollectStats.lens is len(episode_batch), the episode's rows IN THE BUFFER (collector.py:203,990-993): an episode
                # that began before a reset_buffer(keep_statistics=True) counts its rows since the reset, its return is whole
                lens.append(int((ptr[i] - ep_idx[i]) % sub + 1))
                rets.append(np.asarray(ep_rew[i], np.float64).copy())
                self.last_obs[i] = envs[i].reset()
        self.collect_step += steps
        self.collect_episode += len(lens)
        return {"n_collected_steps": steps, "n_collected_episodes": len(lens), "lens": np.asarray(lens, np.int64),
                "returns": np.asarray(rets, np.float64)}


def run_baseline(n_env=64, n_agent=3, horizon=25, minibatch=4096, repeat=1, dispatch="per_agent", budget_s=15.0,
                 seed=1626, ref_default_leg=False):
    torch.manual_seed(seed)
    np.random.seed(seed)
    torch.set_num_threads(min(16, os.cpu_count() or 1))  # the box's CPU share for one GPU; tiny GEMMs do not scale further
    N, D, A, T = n_agent, 6 * n_agent, 5, horizon
    envs = [mpe_oracle.SimpleSpreadWorld(N, T, 0.5, seed + i) for i in range(n_env)]
    actor, critic = _mlp(D, 64, A), _mlp(D, 64, 1)
    opt = torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()), lr=3e-4)
    S = T
    st = PortStores(n_env, S, N, (D,))
    obs_buf, obs_next_buf, act_buf = st.obs, st.obs_next, st.act
    rew_buf, term_buf, trunc_buf, index = st.rew, st.term, st.trunc, st.index
    col = PortCollector(envs, st)

    def act_fn(obs):
        with torch.no_grad():
            logits = actor(torch.from_numpy(obs.reshape(n_env * N, D)))
            return torch.distributions.Categorical(logits=logits).sample().numpy().reshape(n_env, N)

    def one_step(minibatch=minibatch, dispatch=dispatch, collect=True):
        if not collect:  # update only, on the rows of the last collect (the reference-default leg below)
            t_c1 = time.perf_counter()
            return (0.0, *update(minibatch, dispatch, t_c1)[1:])
        t_c0 = time.perf_counter()
        index.reset(keep_statistics=True)
        col.collect(act_fn, n_env * T)
        t_c1 = time.perf_counter()
        return (t_c1 - t_c0, *update(minibatch, dispatch, t_c1)[1:])

    def update(minibatch, dispatch, t_c1):
        idx = index.sample_indices_all()                                   # env-major, time-ordered
        ob, obn = torch.from_numpy(obs_buf[idx]), torch.from_numpy(obs_next_buf[idx])
        with torch.no_grad():
            chunks = lambda x: torch.split(x.reshape(-1, D), 256)          # noqa: E731  a2c.py:122
            v_s = torch.cat([critic(c) for c in chunks(ob)]).reshape(len(idx), N).numpy()
            v_n = torch.cat([critic(c) for c in chunks(obn)]).reshape(len(idx), N).numpy()
        # lanes: [T, n_env*N] time-major view of the env-major flat batch
        tl = lambda x: np.ascontiguousarray(x.reshape(n_env, T, N).transpose(1, 0, 2).reshape(T, n_env * N))  # noqa: E731
        ret, adv = oracle.gae_lanes(tl(v_s), tl(v_n), tl(rew_buf[idx]).astype(np.float32), tl(term_buf[idx]),
                                    tl(trunc_buf[idx]), 0.99, 0.95, threads=1)
        back = lambda x: x.reshape(T, n_env, N).transpose(1, 0, 2).reshape(len(idx), N)  # noqa: E731
        ret, adv = back(ret).astype(np.float32), back(adv).astype(np.float32)
        act_t = torch.from_numpy(act_buf[idx])
        with torch.no_grad():
            logp_old = torch.distributions.Categorical(logits=actor(ob.reshape(-1, D))).log_prob(act_t.reshape(-1)).reshape(len(idx), N)
        groups = [[a] for a in range(N)] if dispatch == "per_agent" else [list(range(N))]
        n_grad = 0
        for g in groups:
            sel = lambda x: torch.as_tensor(x)[:, g].reshape(-1, *x.shape[2:])  # noqa: E731
            o_g, a_g, lp_g = sel(ob), sel(act_t), sel(logp_old)
            adv_g, ret_g = sel(adv), sel(ret)
            n = o_g.shape[0]
            for _ in range(repeat):
                perm = np.random.permutation(n)
                for lo, hi in oracle.split_bounds(n, minibatch, True):
                    mb = torch.as_tensor(perm[lo:hi])
                    dist = torch.distributions.Categorical(logits=actor(o_g[mb]))
                    a_mb = adv_g[mb]
                    a_mb = (a_mb - a_mb.mean()) / (a_mb.std() + 1e-8)
                    ratio = (dist.log_prob(a_g[mb]) - lp_g[mb]).exp()
                    clip_loss = -torch.min(ratio * a_mb, ratio.clamp(0.8, 1.2) * a_mb).mean()
                    vf_loss = (ret_g[mb] - critic(o_g[mb]).flatten()).pow(2).mean()
                    ent = dist.entropy().mean()
                    loss = clip_loss + 0.5 * vf_loss - 0.01 * ent
                    opt.zero_grad()
                    loss.backward()
                    opt.step()
                    _ = (clip_loss.item(), vf_loss.item(), ent.item(), loss.item())
                    n_grad += 1
        t_u1 = time.perf_counter()
        return 0.0, t_u1 - t_c1, n_grad

    one_step()  # warm-up (allocator, torch thread pool)
    t_col = t_upd = 0.0
    n_steps = 0
    t0 = time.perf_counter()
    while True:
        c, u, n_grad = one_step()
        t_col += c
        t_upd += u
        n_steps += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    per_step = (t_col + t_upd) / n_steps
    extra = {}
    if ref_default_leg:
        # the reference's own defaults for the update (trainer.py:287,295: batch_size=64, repeat=1) on the same rows,
        # through the dispatcher's per-agent sequence: ONE timed update (thousands of 64-row gradient steps)
        _, u64, n64 = one_step(minibatch=64, dispatch=dispatch, collect=False)
        extra["update_reference_default_batch64"] = {"ms": u64 * 1e3, "gradient_steps": n64, "batch_size": 64, "repeat": repeat,
                                                     "rows": n_env * T * N, "updates_timed": 1}
    return {
        **extra,
        "value": n_env * N * T / per_step, "unit": "env-steps/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{n_steps} steps of n_env={n_env}, N={N}, T={T}, minibatch={minibatch}, "
                  f"repeat={repeat}, dispatch={dispatch}; python per-env loop + torch-CPU MLP/Adam + single-thread C GAE",
        "collect_env_steps_per_s": n_env * N * T / (t_col / n_steps), "ppo_update_ms": t_upd / n_steps * 1e3,
        "gradient_steps_per_update": n_grad, "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
        "torch_threads": torch.get_num_threads(),
    }


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def gae_legs(shapes=((25, 3072), (25, 32768)), budget_s=3.0, seed=0):
    """SURVEY 8(d) CPU legs for the scan alone, on the layouts the GPU kernel is timed on:
      serial     -- one thread walks every lane's series: the faithful analogue of the reference's single-core numba
                    `_gae` (algorithm_base.py:1079-1134);
      all_cores  -- the same C loop with OpenMP over lanes on the host cores of one GPU's share (16; baseline-best).
    22 B per (lane, step) as for the GPU roofline (f32 in, f64 accumulate)."""
    rng = np.random.default_rng(seed)
    out = {}
    n_thr = min(16, os.cpu_count() or 1)  # the box's CPU share for one GPU (more threads than that only add fork/join cost)
    for T, L in shapes:
        v_s, v_n, rew = (rng.standard_normal((T, L)).astype(np.float32) for _ in range(3))
        term = rng.random((T, L)) < 0.01
        trunc = np.zeros((T, L), bool)
        trunc[-1] = True
        for name, thr in (("serial", 1), ("all_cores", n_thr)):
            oracle.gae_lanes(v_s, v_n, rew, term, trunc, 0.99, 0.95, threads=thr)  # warm-up
            n, t0 = 0, time.perf_counter()
            while True:
                oracle.gae_lanes(v_s, v_n, rew, term, trunc, 0.99, 0.95, threads=thr)
                n += 1
                if time.perf_counter() - t0 >= budget_s / (2 * len(shapes)):
                    break
            dt = (time.perf_counter() - t0) / n
            out[f"gae_{name}_T{T}_L{L}"] = {"us": dt * 1e6, "GBps": 22.0 * T * L / dt / 1e9, "threads": thr, "calls_timed": n}
    return out


if __name__ == "__main__":
    import json

    print(json.dumps(run_baseline(budget_s=5.0)))
