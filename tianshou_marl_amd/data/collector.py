"""Collector -- rollout loop: policy.forward -> env.step -> buffer.add -> reset finished envs -> stats.

API mirror of `tianshou.data.Collector` (/root/reference/tianshou/data/collector.py:257-1098):
`Collector(policy, env, buffer)`, `reset(...)`, `reset_env()`, `reset_buffer(keep_statistics)`,
`collect(n_step=... | n_episode=..., random, reset_before_collect) -> CollectStats`, counters
`collect_step / collect_episode / collect_time`.  Two execution paths, chosen by capability:

* device path -- env has `step_device`, policy has `act_device`, buffer is a DeviceVectorReplayBuffer:
  three HIP launches per vector step (fused actor+critic+sampling, batched env step, index algebra +
  SoA scatter); observations, actions, rewards and flags never leave HBM and there is ONE host
  synchronisation per collect() (to read the episode statistics), versus the reference's per-step
  device<->host copies (collector.py:736, utils/net/common.py:175).
* host path -- any numpy vector env + any policy object: the reference's loop structure incl. the
  ready-env bookkeeping and surplus-env removal for n_episode (collector.py:811-823,1046-1064).
  This is plumbing for config 1 (4 envs) and for third-party envs; rows still land in the device buffer.
"""
from __future__ import annotations

import time
import warnings
from copy import copy
from typing import Any, Callable

import numpy as np
import torch

from .. import ops
from .batch import Batch
from .buffer import DeviceVectorReplayBuffer
from .stats import CollectStats


class Collector:
    def __init__(self, policy, env, buffer: DeviceVectorReplayBuffer | None = None, exploration_noise: bool = False,
                 on_episode_done_hook: Callable | None = None, on_step_hook: Callable | None = None,
                 raise_on_nan_in_buffer: bool = False, use_graph: bool = True, fused_rollout: bool = True) -> None:
        self.use_graph = use_graph
        self.fused_rollout = fused_rollout
        self.policy = getattr(policy, "policy", policy)  # an Algorithm is accepted (collector.py:358)
        self.env = env
        self.env_num = len(env)
        if buffer is None:
            raise ValueError("a DeviceVectorReplayBuffer is required (the buffer lives in HBM)")
        if buffer.buffer_num < self.env_num:  # _validate_buffer, collector.py:369-389
            raise ValueError(f"Buffer has only {buffer.buffer_num} sub-buffers for {self.env_num} envs")
        self.buffer = buffer
        self.exploration_noise = exploration_noise
        self.on_episode_done_hook, self.on_step_hook = on_episode_done_hook, on_step_hook
        self.raise_on_nan_in_buffer = raise_on_nan_in_buffer
        self.collect_step = self.collect_episode = 0
        self.collect_time = 0.0
        self._pre_obs = None
        self._pre_info = None
        self._device_path = hasattr(env, "step_device") and hasattr(self.policy, "act_device")
        self._ws: dict = {}

    # ---- resets (collector.py:392-459) --------------------------------------------------------------
    def reset(self, reset_buffer: bool = True, reset_stats: bool = True, gym_reset_kwargs: dict | None = None):
        self.reset_env(gym_reset_kwargs)
        if reset_buffer:
            self.reset_buffer()
        if reset_stats:
            self.reset_stat()
        return self._pre_obs, self._pre_info

    def reset_stat(self) -> None:
        self.collect_step = self.collect_episode = 0
        self.collect_time = 0.0

    def reset_buffer(self, keep_statistics: bool = False) -> None:
        self.buffer.reset(keep_statistics=keep_statistics)

    def reset_env(self, gym_reset_kwargs: dict | None = None) -> None:
        if self._device_path:
            self._pre_obs = self.env.reset_device()
            self._pre_info = None
        else:
            self._pre_obs, self._pre_info = self.env.reset(**(gym_reset_kwargs or {}))

    # ---- collect -----------------------------------------------------------------------------------
    def collect(self, n_step: int | None = None, n_episode: int | None = None, random: bool = False,
                render: float | None = None, reset_before_collect: bool = False,
                gym_reset_kwargs: dict | None = None) -> CollectStats:
        # input validation of collector.py:461-497
        if n_step is not None and n_episode is not None:
            raise ValueError(f"Only one of n_step or n_episode is allowed in Collector.collect, got {n_step=}, {n_episode=}.")
        if n_step is None and n_episode is None:
            raise ValueError("Either n_step or n_episode should be set (and > 0)")
        if n_step is not None:
            if n_step <= 0:
                raise ValueError(f"n_step should be > 0, got {n_step}")
            if n_step % self.env_num != 0:
                warnings.warn(f"{n_step=} is not a multiple of ({self.env_num=}), which may cause extra transitions "
                              "being collected into the buffer.", stacklevel=2)
        elif n_episode <= 0:
            raise ValueError(f"n_episode should be > 0, got {n_episode}")
        if reset_before_collect:
            self.reset(reset_buffer=False, gym_reset_kwargs=gym_reset_kwargs)
        if self._pre_obs is None:
            raise ValueError("Initial obs and info should not be None. Either reset the collector (using reset or "
                             "reset_env) or pass reset_before_collect=True to collect.")
        t0 = time.time()
        prev_training = getattr(self.policy, "training", None)
        if prev_training is not None and hasattr(self.policy, "train"):
            self.policy.train(False)  # torch_train_mode(policy, False), collector.py:500
        try:
            if hasattr(self.policy, "net") and hasattr(self.policy.net, "sync_image"):
                self.policy.net.sync_image()
            if self._device_path and not random:
                stats = self._collect_device(n_step, n_episode)
            else:
                stats = self._collect_host(n_step, n_episode, random, gym_reset_kwargs)
        finally:
            if prev_training is not None and hasattr(self.policy, "train"):
                self.policy.train(prev_training)
        if self.raise_on_nan_in_buffer and self.buffer.hasnull():
            from .._abi import MalformedBufferError

            raise MalformedBufferError("NaN detected in the buffer.")
        dt = max(time.time() - t0, 1e-9)
        stats.set_collect_time(dt)
        self.collect_time += dt
        return stats

    # ---- device path -----------------------------------------------------------------------------------
    def _device_ws(self, n_iter: int):
        key = ("dev", n_iter)
        if key not in self._ws:
            E, N, dev = self.env_num, self.buffer.n_agent, self.buffer.device
            self._ws[key] = dict(
                act=torch.zeros(E, N, dtype=torch.int32, device=dev), logp=torch.zeros(E, N, device=dev),
                value=torch.zeros(E, N, device=dev),
                ptr=torch.zeros(n_iter, E, dtype=torch.int64, device=dev),
                ep_idx=torch.zeros(n_iter, E, dtype=torch.int64, device=dev),
                ep_len=torch.zeros(n_iter, E, dtype=torch.int64, device=dev),
                ep_rew=torch.zeros(n_iter, E, N, dtype=torch.float64, device=dev))
        return self._ws[key]

    def _collect_device(self, n_step: int | None, n_episode: int | None) -> CollectStats:
        env, buf, pol = self.env, self.buffer, self.policy
        E, N = self.env_num, buf.n_agent
        if n_step is not None:
            n_iter = -(-n_step // E)
            ws = self._device_ws(n_iter)
            out = dict(act=ws["act"].view(-1), logp=ws["logp"].view(-1), value=ws["value"].view(-1), logits=None)

            def body():
                pp0 = env._pp
                for it in range(n_iter):
                    obs = env.obs_cur  # stays intact: the env writes the next policy input into its other buffer
                    # sampling counter lives in HBM (env.rng_tick) so that a captured graph advances it on replay
                    pol.act_device(obs, out=out, offset_dev=env.rng_tick)
                    obs_next, rew, term, trunc, done = env.step_device(ws["act"], rng_tick_inc=E * N)
                    buf.add_device(obs, ws["act"], rew, term, trunc, obs_next, ws["logp"], ws["value"], None, done,
                                   outs=(ws["ptr"][it], ws["ep_rew"][it], ws["ep_len"][it], ws["ep_idx"][it]))
                if env._pp != pp0:  # odd number of ping-pong flips: restore the parity the graph was captured with
                    env._obs_pp[pp0].copy_(env.obs_cur)
                    env._pp = pp0

            mode = bool(getattr(pol, "deterministic_eval", False) and not getattr(pol, "is_within_training_step", False))
            gkey = ("graph", n_iter, mode, env._pp)
            if self._can_fuse():
                self._rollout_fused(n_iter, ws, mode)  # the whole loop in ONE persistent kernel (csrc/rollout.hip)
            elif self.use_graph and self._ws.get(("seen", n_iter)):
                if gkey not in self._ws:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        body()
                    self._ws[gkey] = g
                self._ws[gkey].replay()
            else:
                body()  # first call: eager (also sets one-time kernel attributes before any capture)
                self._ws[("seen", n_iter)] = True
            lens = ws["ep_len"][:n_iter]
            mask = lens > 0
            lens_h = lens[mask].cpu().numpy()  # the single host sync of this collect()
            rets_h = ws["ep_rew"][:n_iter][mask].cpu().numpy()
            n_ep, steps = int(mask.sum().item()), n_iter * E
            self._pre_obs = env.obs_cur
        else:
            return self._collect_device_episodes(n_episode)
        self.collect_step += steps
        self.collect_episode += n_ep
        return CollectStats.with_autogenerated_stats(returns=rets_h, lens=lens_h, n_collected_episodes=n_ep,
                                                     n_collected_steps=steps)

    def _can_fuse(self) -> bool:
        env, buf, pol = self.env, self.buffer, self.policy
        return bool(self.fused_rollout and hasattr(env, "_cfg") and getattr(env, "n_act", 0) == 5
                    and hasattr(pol, "net") and getattr(pol.net, "hidden", 0) == 64 and buf.vnext_store is not None
                    and buf.obs_next_store is not None and buf.buffer_num == self.env_num
                    and self.on_step_hook is None and self.on_episode_done_hook is None)

    def _rollout_fused(self, n_iter: int, ws: dict, deterministic: bool) -> None:
        import ctypes as C

        from .._abi import call, ptr, stream_ptr, tsm_rollout_desc

        env, buf, pol = self.env, self.buffer, self.policy
        d = tsm_rollout_desc()
        d.params, d.param_image = ptr(pol.net.flat.data), ptr(pol.net.image)
        d.obs_dim, d.hidden, d.n_act, d.mode = env.obs_dim, pol.net.hidden, env.n_act, 2 if deterministic else 1
        d.policy_seed, d.offset, d.offset_dev = pol.seed & (2**64 - 1), 0, ptr(env.rng_tick)
        d.env, d.env_seed, d.episode_ctr = env._cfg, env._seed & (2**64 - 1), ptr(env.episode_ctr)
        d.agent_pos, d.agent_vel, d.landmark_pos = ptr(env.agent_pos), ptr(env.agent_vel), ptr(env.landmark_pos)
        d.steps, d.auto_reset, d.n_steps = ptr(env.steps), int(env.auto_reset), n_iter
        d.obs_cur_out = ptr(env.obs_cur)
        d.vrb_state, d.sub_size, d.done_store = ptr(buf.index.state), buf.sub_size, ptr(buf.done_store)
        d.obs_store, d.obs_next_store, d.rew_store = ptr(buf.obs_store), ptr(buf.obs_next_store), ptr(buf.rew_store)
        d.logp_store, d.vs_store, d.vnext_store = ptr(buf.logp_store), ptr(buf.vs_store), ptr(buf.vnext_store)
        d.act_store, d.term_store, d.trunc_store = ptr(buf.act_store), ptr(buf.term_store), ptr(buf.trunc_store)
        d.ptr_out, d.ep_rew_out = ptr(ws["ptr"]), ptr(ws["ep_rew"])
        d.ep_len_out, d.ep_idx_out = ptr(ws["ep_len"]), ptr(ws["ep_idx"])
        s = stream_ptr()
        call("tsm_rollout_spread", C.byref(d), s)
        call("tsm_u64_add", ptr(env.rng_tick), n_iter * self.env_num * buf.n_agent, s)
        buf.mark_policy_outputs(getattr(pol, "param_version", -1))

    def _collect_device_episodes(self, n_episode: int) -> CollectStats:
        env, buf, pol = self.env, self.buffer, self.policy
        E, N, dev = self.env_num, buf.n_agent, buf.device
        ready = torch.arange(min(E, n_episode), device=dev)
        ws = self._device_ws(1)
        out = dict(act=ws["act"].view(-1), logp=ws["logp"].view(-1), value=ws["value"].view(-1), logits=None)
        lens, rets, steps, n_ep = [], [], 0, 0
        while True:
            obs_prev = env.obs_cur
            pol.act_device(obs_prev, out=out)
            obs_next, rew, term, trunc, done = env.step_device(ws["act"])
            full = ready.numel() == E
            g = (lambda x: x) if full else (lambda x: ops.gather_rows(x, ready))
            _, ep_rew, ep_len, _ = buf.add_device(g(obs_prev), g(ws["act"]), g(rew), g(term), g(trunc), g(obs_next),
                                                  g(ws["logp"]), g(ws["value"]), None if full else ready,
                                                  g(done.view(-1, 1)).view(-1))
            steps += ready.numel()
            d = ep_len > 0
            nd = int(d.sum().item())
            if nd:
                lens.append(ep_len[d].cpu().numpy())
                rets.append(ep_rew[d].cpu().numpy())
                n_ep += nd
                surplus = ready.numel() - (n_episode - n_ep)  # collector.py:1046-1064
                if 0 < surplus and n_ep < n_episode:
                    drop = torch.nonzero(d).view(-1)[:surplus]
                    keep = torch.ones(ready.numel(), dtype=torch.bool, device=dev)
                    keep[drop] = False
                    ready = ready[keep]
            if n_ep >= n_episode:
                break
        self.collect_step += steps
        self.collect_episode += n_ep
        self.reset_env()  # collector.py:1095-1097
        return CollectStats.with_autogenerated_stats(
            returns=np.concatenate(rets) if rets else np.array([]), lens=np.concatenate(lens) if lens else np.array([], int),
            n_collected_episodes=n_ep, n_collected_steps=steps)

    # ---- host path (reference loop structure) -----------------------------------------------------------
    def _policy_act(self, obs, info, random: bool, ready: np.ndarray):
        if random:
            spaces = self.env.action_space
            act = np.array([spaces[i].sample() for i in ready]) if isinstance(spaces, list) else np.array(
                [spaces.sample() for _ in ready])
            return act, Batch()
        with torch.no_grad():
            res = self.policy(Batch(obs=obs, info=info), None)
        act = res.act
        act = act.detach().cpu().numpy() if isinstance(act, torch.Tensor) else np.asarray(act)
        pol = res.policy if "policy" in res else Batch()
        return act, pol

    def _collect_host(self, n_step, n_episode, random, gym_reset_kwargs) -> CollectStats:
        ready = np.arange(self.env_num) if n_step is not None else np.arange(min(self.env_num, n_episode))
        last_obs = self._pre_obs[ready] if len(ready) != self.env_num else self._pre_obs
        last_info = None if self._pre_info is None else self._pre_info[ready]
        step_count = n_ep = 0
        ep_rets, ep_lens = [], []
        while True:
            act, pol = self._policy_act(last_obs, last_info, random, ready)
            obs_next, rew, term, trunc, info = self.env.step(act, ready)
            term_e = np.asarray(term).reshape(len(ready), -1).any(1)
            trunc_e = np.asarray(trunc).reshape(len(ready), -1).any(1)
            done = term_e | trunc_e  # env-level reduction of per-agent lists (quirk Q4)
            step_batch = Batch(obs=last_obs, act=act, rew=rew, terminated=term, truncated=trunc, obs_next=obs_next,
                               info=info)
            if len(pol.get_keys()):
                step_batch.policy = pol
            if self.on_step_hook is not None:
                self.on_step_hook(step_batch)
            step_count += len(ready)
            n_done = int(done.sum())
            n_ep += n_done
            ptr, ep_rew, ep_len, ep_idx = self.buffer.add(step_batch, buffer_ids=ready)
            last_obs, last_info = copy(obs_next), copy(info)
            if n_done:
                d_local = np.where(done)[0]
                ep_lens.extend(ep_len[d_local])
                ep_rets.extend(ep_rew[d_local])
                obs_reset, info_reset = self.env.reset(env_id=ready[d_local], **(gym_reset_kwargs or {}))
                if self.on_episode_done_hook is not None:
                    for li in d_local:
                        idx = self.buffer.get_buffer_indices(int(ep_idx[li]), int(ptr[li] + 1))
                        self.on_episode_done_hook(self.buffer[idx])
                last_obs[d_local] = obs_reset
                last_info[d_local] = info_reset
                if n_episode:
                    surplus = len(ready) - (n_episode - n_ep)
                    if surplus > 0:
                        keep = np.ones(len(ready), bool)
                        keep[d_local[:surplus]] = False
                        ready, last_obs, last_info = ready[keep], last_obs[keep], last_info[keep]
            if (n_step and step_count >= n_step) or (n_episode and n_ep >= n_episode):
                break
        self.collect_step += step_count
        self.collect_episode += n_ep
        if n_step:
            self._pre_obs, self._pre_info = last_obs, last_info
        else:
            self.reset_env(gym_reset_kwargs)
        return CollectStats.with_autogenerated_stats(
            returns=np.array(ep_rets, dtype=float), lens=np.array(ep_lens, dtype=int), n_collected_episodes=n_ep,
            n_collected_steps=step_count)


_ = Any
