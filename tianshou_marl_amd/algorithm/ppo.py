"""PPO on the device-resident rollout: policy forward, GAE, clip loss, backward and Adam in HIP.

API mirror of the reference's on-policy actor-critic stack:
  `PPO(...)`                       tianshou/algorithm/modelfree/ppo.py:17-224 (same hyper-parameter names/defaults)
  `policy(batch, state)`           ProbabilisticActorPolicy.forward, modelfree/reinforce.py:167-192
  `update(buffer, batch_size, repeat)`   OnPolicyAlgorithm.update -> Algorithm._update, algorithm_base.py:584-629,852-863
  `_preprocess_batch`              a2c.py:113-151 + ppo.py:146-162 (critic passes, GAE, logp_old)
  `_update_with_batch`             ppo.py:164-224 (minibatch loop, Batch.split merge_last rule, 4 loss statistics)
  `Optimizer.step`                 algorithm_base.py:485-498 (clip_grad_norm_ + Adam)
  `learn(batch)`                   the `.learn()` entry the MARL trainers call (training_coordinator.py:336)
(all paths relative to /root/reference).  Per-agent dispatch of a shared algorithm object
(MARLDispatcher, multiagent/marl.py:208-268) is selected with `dispatch="per_agent"`.
"""
from __future__ import annotations

import os
import time
from typing import Literal

import numpy as np
import torch
from torch import nn

from .. import ops
from ..data.batch import Batch, split_bounds
from ..data.buffer import DeviceVectorReplayBuffer
from ..data.stats import A2CTrainingStats, LazyStats, MapTrainingStats, SequenceSummaryStats
from ..utils.net import DeviceRunningMeanStd, DiscreteActorCritic


class PPO(nn.Module):
    def __new__(cls, *args, **kw):
        # reference-style construction `PPO(policy=..., critic=..., optim=...)`: nets the fused 64-wide kernels do not
        # cover (other widths / depths, tanh, a wider critic input) run on the general kernels of GenericPPO
        if cls is PPO and kw.get("policy") is not None and kw.get("net") is None:
            from ..utils.net import MLPActorCritic, net_from_reference_modules

            if isinstance(net_from_reference_modules(kw["policy"], kw.get("critic"), kw.get("device", "cuda")), MLPActorCritic):
                from .ppo_generic import GenericPPO

                return super().__new__(GenericPPO)
        return super().__new__(cls)

    def __init__(self, *, policy=None, critic=None, optim=None, eps_clip: float = 0.2, dual_clip: float | None = None,
                 value_clip: bool = False, advantage_normalization: bool = True, recompute_advantage: bool = False,
                 vf_coef: float = 0.5, ent_coef: float = 0.01, max_grad_norm: float | None = None,
                 gae_lambda: float = 0.95, max_batchsize: int = 256, gamma: float = 0.99, return_scaling: bool = False,
                 net: DiscreteActorCritic | None = None, lr: float = 3e-4, betas=(0.9, 0.999), adam_eps: float = 1e-8,
                 weight_decay: float = 0.0, deterministic_eval: bool = False,
                 dispatch: Literal["per_agent", "pooled"] | None = None,
                 shuffle: Literal["numpy", "device"] = "numpy", seed: int = 0, use_graph: bool = True,
                 async_stats: bool = False, device: str | torch.device = "cuda") -> None:
        """Either the reference's arguments (ppo.py:17-60: `policy` = DiscreteActorPolicy, `critic` = DiscreteCritic,
        `optim` = AdamOptimizerFactory -- utils/ref_nets.py, algorithm/optim.py) or the engine's own (`net` = a flat
        HBM network + Adam hyper-parameters).  Hyper-parameter names and defaults are the reference's."""
        super().__init__()
        assert dual_clip is None or dual_clip > 1.0, f"Dual-clip PPO parameter should greater than 1.0 but got {dual_clip}"
        assert 0.0 <= gae_lambda <= 1.0, f"GAE lambda should be in [0, 1] but got: {gae_lambda}"
        sched_factory = None
        if net is None:
            if policy is None:
                raise TypeError("PPO needs `policy=` (+ `critic=`, `optim=`) or `net=`")
            from ..utils.net import net_from_reference_modules

            net = net_from_reference_modules(policy, critic, device)
            deterministic_eval = bool(getattr(policy, "deterministic_eval", deterministic_eval))
            # a reference PPO object updates on the whole batch it is handed (ppo.py:164-224); the per-agent sequence of
            # MARLDispatcher (marl.py:251-268) is what `dispatch="per_agent"` reproduces for a shared algorithm
            dispatch = dispatch or "pooled"
        dispatch = dispatch or "per_agent"
        if optim is not None:  # AdamOptimizerFactory (optim.py:91-111), optionally with an LR scheduler factory
            kw_o = optim.adam_kwargs()
            lr, betas, adam_eps, weight_decay = kw_o["lr"], kw_o["betas"], kw_o["adam_eps"], kw_o["weight_decay"]
            sched_factory = getattr(optim, "lr_scheduler_factory", None)
        self._ctor = dict(lr=lr, betas=betas, adam_eps=adam_eps, weight_decay=weight_decay, eps_clip=eps_clip,
                          dual_clip=dual_clip, value_clip=value_clip, advantage_normalization=advantage_normalization,
                          recompute_advantage=recompute_advantage, vf_coef=vf_coef, ent_coef=ent_coef,
                          max_grad_norm=max_grad_norm, gae_lambda=gae_lambda, max_batchsize=max_batchsize, gamma=gamma,
                          return_scaling=return_scaling, deterministic_eval=deterministic_eval, dispatch=dispatch,
                          shuffle=shuffle, seed=seed, use_graph=use_graph, async_stats=async_stats)
        self._sched_factory = sched_factory
        self.net = net
        # Policy attributes the collector / MARL containers read (algorithm_base.py:159-373, marl.py:79-85)
        from ..env.spaces import Box, Discrete

        self.action_space = Discrete(net.n_act)
        self.observation_space = Box(-np.inf, np.inf, (net.obs_dim,))
        # the learning rate lives in HBM too (`_lr_dev`): LR schedulers move it between updates and the captured update
        # graph reads it in place (algorithm_base.py:626-627)
        self._lr_dev = torch.tensor([float(lr)], dtype=torch.float64, device=net.flat.device)
        self._lr = float(lr)
        self.lr_schedulers: list = []
        if sched_factory is not None:  # Algorithm._create_optimizer (algorithm_base.py:506-519)
            self.lr_schedulers.append(sched_factory.create_scheduler(self))
        self.betas, self.adam_eps, self.weight_decay = betas, adam_eps, weight_decay
        self.eps_clip, self.dual_clip, self.value_clip = eps_clip, dual_clip, value_clip
        self.advantage_normalization, self.recompute_adv = advantage_normalization, recompute_advantage
        self.vf_coef, self.ent_coef, self.max_grad_norm = vf_coef, ent_coef, max_grad_norm
        self.gae_lambda, self.gamma, self.max_batchsize = gae_lambda, gamma, max_batchsize
        self.return_scaling, self.ret_rms, self._eps = return_scaling, DeviceRunningMeanStd(net.flat.device), 1e-8
        self.deterministic_eval = deterministic_eval
        self.reuse_rollout_outputs = True  # learn(): take logp_old / V(obs) / V(obs_next) the rollout stored (_stored_outputs_ok)
        self.is_within_training_step = False
        self.dispatch, self.shuffle = dispatch, shuffle
        self.use_graph = use_graph
        self.async_stats = async_stats
        self.seed = int(seed)
        dev = net.flat.device
        self.exp_avg = torch.zeros_like(net.flat.data)
        self.exp_avg_sq = torch.zeros_like(net.flat.data)
        self.opt_step = 0
        self._adam_work = torch.zeros(ops.call("tsm_adam_work_elems", net.flat.numel()), dtype=torch.float32, device=dev)
        self._sample_ctr = 0  # Philox counter base for action sampling
        self._perm_ctr = torch.zeros(1, dtype=torch.int64, device=dev)  # draw counter of the device-side permutations
        self._cfg = ops.make_ppo_cfg(eps_clip, dual_clip, value_clip, advantage_normalization, vf_coef, ent_coef)
        self._ws: dict = {}
        self._grad_sync = None  # set by parallel.attach_data_parallel
        # N>1: capture the per-step gradient all-reduce (RCCL) inside the update hipGraph (TSM_GRAPH_COLLECTIVES=0: eager)
        self.graph_collectives = os.environ.get("TSM_GRAPH_COLLECTIVES", "1") != "0"
        self.param_version = 0  # bumped whenever the parameters change (stored rollout outputs become stale)

    # the reference collector accepts an Algorithm and uses `.policy` (collector.py:358)
    @property
    def policy(self) -> "PPO":
        return self

    @property
    def device(self) -> torch.device:
        return self.net.flat.device

    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, value: float) -> None:
        self._lr = float(value)
        self._lr_dev.fill_(self._lr)

    # ---- rollout side -------------------------------------------------------------------------
    def act_device(self, obs: torch.Tensor, out: dict | None = None, offset_dev: torch.Tensor | None = None,
                   row_offset: int = 0) -> dict:
        """obs [..., D] in HBM -> dict(act i32, logp, value) for every row (one fused kernel).
        `row_offset` shifts the sampling counter: callers that serve several agents with one policy in separate calls
        (MultiAgentPolicy) give each call its own counter range so that the draws are independent."""
        rows = obs.reshape(-1, self.net.obs_dim)
        mode = "mode" if (self.deterministic_eval and not self.is_within_training_step) else "sample"
        res = ops.policy_forward(self.net.flat.data, rows, self.net.n_act, self.net.hidden, image=self.net.image, mode=mode, seed=self.seed,
                                 offset=self._sample_ctr + row_offset, want_logits=out is None, offset_dev=offset_dev, out=out)
        if offset_dev is None:
            self._sample_ctr += rows.shape[0]
        return res

    def forward(self, batch: Batch, state=None, **kwargs) -> Batch:
        """reinforce.py:167-192: Batch(logits, act, state, ...) for `batch.obs` (array [B, D] or Batch(obs=...))."""
        obs = batch.obs
        if isinstance(obs, Batch) and "observations" in obs:  # parallel-mode joint rows -> [R, N, D] in agent order
            from ..data.buffer import _obs_array

            obs = _obs_array(obs)
        elif isinstance(obs, Batch) and "obs" in obs:  # MARL wrappers hand over obs.obs (marl.py:157-161)
            obs = obs.obs
        obs_t = torch.as_tensor(np.asarray(obs) if not isinstance(obs, torch.Tensor) else obs).to(self.device, torch.float32)
        lead = obs_t.shape[:-1]
        res = self.act_device(obs_t)
        return Batch(logits=res["logits"].reshape(*lead, -1), act=res["act"].reshape(lead).to(torch.int64),
                     state=None, policy=Batch(logp=res["logp"].reshape(lead), v_s=res["value"].reshape(lead)))

    def map_action(self, act):
        return act  # discrete: identity (algorithm_base.py:255-289)

    def map_action_inverse(self, act):
        return act

    def add_exploration_noise(self, act, batch):
        return act

    # ---- update side --------------------------------------------------------------------------
    def _valid_rows(self, buffer: DeviceVectorReplayBuffer):
        """Time-major joint rows (slot*B + env) that hold data, plus per-env ragged descriptors."""
        lengths = buffer.index.lengths
        lens_h = lengths.cpu().numpy()
        S, B = buffer.sub_size, buffer.buffer_num
        uniform = bool((lens_h == lens_h[0]).all())
        ins_h = buffer.index.insertion_idx.cpu().numpy()
        start_h = (ins_h - lens_h) % S
        if uniform and (start_h == 0).all():
            T = int(lens_h[0])
            return T, None, None, None
        # joint-row ids (slot * B + env) in the reference's sample(0) order: env-major, time-ordered inside a
        # sub-buffer (manager.py:224-229, buffer_base.py:511-514)
        k = np.arange(S)[None, :]
        slot = (start_h[:, None] + k) % S
        valid = k < lens_h[:, None]
        rows = (slot * B + np.arange(B)[:, None])[valid]
        dev = self.device
        return S, torch.as_tensor(rows).to(dev), torch.as_tensor(start_h.astype(np.int32)).to(dev), \
            torch.as_tensor(lens_h.astype(np.int32)).to(dev)

    def _preprocess_batch(self, buffer: DeviceVectorReplayBuffer) -> dict:
        """a2c.py:113-151 + ppo.py:146-162 on the whole buffer (sample(0) == every stored row)."""
        T, rows, env_start, env_len = self._valid_rows(buffer)
        B, N, D = buffer.buffer_num, buffer.n_agent, buffer.obs_dim
        L = B * N
        P = self.net.flat.data
        obs = buffer.obs_store[:T].reshape(T * L, D)
        act = buffer.act_store[:T].reshape(T * L)
        # critic(obs), critic(obs_next), logp_old: two fused passes (no max_batchsize chunking needed in HBM)
        cur = ops.policy_forward(P, obs, self.net.n_act, self.net.hidden, image=self.net.image, mode="given", act=act, want_logits=False)
        v_s, logp_old = cur["value"].view(T, L), cur["logp"]
        if buffer.obs_next_store is None:  # ignore_obs_next: obs_next is obs[next(index)] (buffer_base.py:612-616)
            v_next = self._next_values_by_index(buffer, v_s, T, rows)
        else:
            v_next = ops.policy_forward(P, buffer.obs_next_store[:T].reshape(T * L, D), self.net.n_act, self.net.hidden,
                                        mode="none", want_logits=False)["value"].view(T, L)
        ret, adv = self._gae(v_s, v_next, buffer.rew_store[:T].reshape(T, L), buffer.term_store[:T].reshape(T, L),
                             buffer.trunc_store[:T].reshape(T, L), N, env_start=env_start, env_len=env_len, rows=rows)
        return dict(T=T, rows=rows, obs=obs, act=act, v_s=v_s.reshape(-1), ret=ret.reshape(-1), adv=adv.reshape(-1),
                    logp_old=logp_old, n_env=B, n_agent=N)

    def _next_values_by_index(self, buffer: DeviceVectorReplayBuffer, v_s: torch.Tensor, T: int, rows, out=None) -> torch.Tensor:
        """V(obs_next) [T, L] for a buffer WITHOUT an obs_next store (`ignore_obs_next=True`): the reference then reads
        obs_next as obs[next(index)] (buffer_base.py:612-616), next(index) being the row itself at an episode end and at the
        newest row -- so the critic pass over obs_next is a re-indexing of the pass over obs.  Uniform, unrotated slots:
        one launch (tsm_value_next_index); ragged / rotated sub-buffers: the buffer's own `next` on the flat indices."""
        B, N = buffer.buffer_num, buffer.n_agent
        if rows is None:
            return ops.value_next_index(v_s, buffer.done_store[:T], T, B * N, N, out=out)
        S, dev = buffer.sub_size, self.device
        flat = (torch.arange(B, device=dev).view(1, B) * S + torch.arange(S, device=dev).view(S, 1)).reshape(-1)
        nxt = buffer.index.next(flat)
        return v_s.view(S, B, N)[nxt % S, torch.div(nxt, S, rounding_mode="floor")].reshape(S, B * N)

    def _gae(self, v_s, v_next, rew, term, trunc, N: int, env_start=None, env_len=None, rows=None, out=None):
        """compute_episodic_return on [T, L] lanes, with the return_scaling arithmetic of a2c.py:132-146 when enabled:
        the critic values are un-normalised by sqrt(ret_rms.var + eps), the returns divided by it, and ret_rms is then
        updated with the unnormalised returns -- all on device (statistics in HBM), so the pass can be captured."""
        if not self.return_scaling:
            return ops.gae_lanes(v_s, v_next, rew, term, trunc, self.gamma, self.gae_lambda, lanes_per_env=N,
                                 env_start=env_start, env_len=env_len, out=out)
        rms = self.ret_rms.dev
        T, L = v_s.shape
        if self.dispatch == "per_agent" and N > 1:
            # MARLDispatcher preprocesses agent after agent with the SAME algorithm object (marl.py:208-249): the running
            # statistics advance between the agents, so agent a + 1 is scaled with what agent a left behind
            B = L // N
            ret, adv = out if out is not None else (torch.empty_like(v_s), torch.empty_like(v_s))
            lane = lambda x, a: x.view(T, B, N)[:, :, a].contiguous()  # noqa: E731
            for a in range(N):
                r, ad = ops.gae_lanes(lane(v_s, a), lane(v_next, a), lane(rew, a), lane(term, a), lane(trunc, a),
                                      self.gamma, self.gae_lambda, lanes_per_env=1, env_start=env_start, env_len=env_len,
                                      rms=rms, rms_eps=self._eps)
                self.ret_rms.update_scaled_returns(r, self._eps, ids=rows)  # rows index the [T, B] lane array
                ret.view(T, B, N)[:, :, a].copy_(r)
                adv.view(T, B, N)[:, :, a].copy_(ad)
            return ret, adv
        ret, adv = ops.gae_lanes(v_s, v_next, rew, term, trunc, self.gamma, self.gae_lambda, lanes_per_env=N,
                                 env_start=env_start, env_len=env_len, out=out, rms=rms, rms_eps=self._eps)
        ids = None if rows is None else (rows[:, None] * N + torch.arange(N, device=rows.device)[None, :]).reshape(-1)
        self.ret_rms.update_scaled_returns(ret, self._eps, ids=ids)
        return ret, adv

    def _sample_ids(self, pb: dict, agent: int | None) -> torch.Tensor | None:
        """Flat sample ids ((slot*B+env)*N + agent) of the rows to train on; None = all, contiguous."""
        T, B, N, rows = pb["T"], pb["n_env"], pb["n_agent"], pb["rows"]
        dev = self.device
        if rows is None and agent is None and self.shuffle != "numpy":
            return None
        if rows is not None:
            base = rows
        elif self.shuffle == "numpy":
            # parity mode: position p of the reference batch (env-major, time-ordered: sample(0)) -> joint row id,
            # so that np.random.permutation picks the very rows Batch.split would (batch.py:1219)
            base = ref_order_rows(T, B, dev)
        else:
            base = torch.arange(T * B, device=dev)
        if agent is None:
            return (base[:, None] * N + torch.arange(N, device=dev)[None, :]).reshape(-1)
        return base * N + agent

    def _global_adv_stats(self, stats, mb_start):
        """Data-parallel replicas normalise with the statistics of the GLOBAL minibatch (the union over the ranks):
        one small all-reduce for all minibatches in `stats` (parallel.GradSync.merge_adv_stats_); no-op on one GPU."""
        gs = self._grad_sync
        if gs is not None and stats is not None and gs.global_adv_stats:
            gs.merge_adv_stats_(stats.view(-1, 2), mb_start)
        return stats

    def _capture_graph(self, graph: "torch.cuda.CUDAGraph", fn) -> None:
        """Capture `fn()` into `graph`.  With data-parallel collectives inside (RCCL): every rank captures the same
        collective sequence, on a side stream in thread-local capture mode (the process group's watchdog thread may touch
        the device meanwhile).  Whether a backend CAN be captured is decided before this point
        (parallel.attach_data_parallel: only RCCL; bench.py probes it in a child process).  A capture that fails all the
        same is fatal: the process group's own stream has joined the capture, HIP leaves it invalidated, and every later
        collective on it dies (hipErrorStreamCaptureInvalidated) -- there is no in-process fall-back; the job is started
        again with TSM_GRAPH_COLLECTIVES=0 (collectives outside the graphs)."""
        if self._grad_sync is None:
            with ops.graph_capture(graph):
                fn()
            return
        dev = self.device
        err = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with ops.gc_hold():  # (no collection inside a capture: see ops.graph_capture)
            with torch.cuda.stream(side):
                graph.capture_begin(capture_error_mode="thread_local")
                try:
                    fn()
                except Exception as e:  # noqa: BLE001
                    err = e
                try:
                    graph.capture_end()
                except Exception as e:  # noqa: BLE001
                    err = err or e
                if err is not None:
                    ops.call("tsm_stream_abort_capture", side.cuda_stream)
        if err is not None:
            raise RuntimeError(
                "capturing the gradient all-reduce into the update hipGraph failed "
                f"({type(err).__name__}: {err}).  Streams behind an invalidated capture are not usable, so this "
                "process cannot continue: start the job again with TSM_GRAPH_COLLECTIVES=0 (collectives outside "
                "the graphs)."
            ) from err
        torch.cuda.current_stream().wait_stream(side)

    def _global_adv_stats_steps(self, stats, mb_start):
        """`_global_adv_stats` for the captured update: pack, YIELD the f64 pack (the caller sums it over the ranks), unpack."""
        gs = self._grad_sync
        if gs is not None and stats is not None and gs.wants_global_adv_stats():
            pack_fn, unpack_fn = gs._stat_codec or gs._default_codec()
            pack = pack_fn(stats, mb_start)
            yield pack
            unpack_fn(pack, stats)

    def _update_with_batch(self, pb: dict, batch_size: int | None, repeat: int, agent: int | None = None,
                           buffer: DeviceVectorReplayBuffer | None = None, perm_base: int | None = None) -> A2CTrainingStats:
        """ppo.py:164-224 for one sample set (all lanes, or one agent's lanes under per-agent dispatch)."""
        return drive_steps(self._update_steps(pb, batch_size, repeat, agent=agent, buffer=buffer, perm_base=perm_base),
                           self._grad_sync)

    def _device_perm(self, n: int, perm_base: int | None, agent: int | None, repeat: int, step: int) -> torch.Tensor:
        """The device-side permutation (shuffle="device") of repeat `step` for `agent`'s sample set.  Inside `update()`
        (`perm_base` = the optimizer step count when the update began) it is draw number perm_base + agent * repeat + step
        of the keyed generator -- exactly the draws the captured update makes in one launch (`_update_graph`), so eager and
        captured updates shuffle alike and a checkpoint continues bit-identically in either mode.  `learn()` calls
        (perm_base None) count their draws in `_perm_ctr`."""
        if perm_base is not None:
            return ops.random_permutations(n, 1, self.seed ^ 0x5DEECE66D, counter=perm_base + (agent or 0) * repeat + step,
                                           device=self.device)[0]
        perm = ops.random_permutations(n, 1, self.seed ^ 0x5DEECE66D, counter_dev=self._perm_ctr, device=self.device)[0]
        ops.call("tsm_u64_add", ops.ptr(self._perm_ctr), 1, ops.stream_ptr())
        return perm

    def _update_steps(self, pb: dict, batch_size: int | None, repeat: int, agent: int | None = None,
                      buffer: DeviceVectorReplayBuffer | None = None, perm_base: int | None = None):
        """The minibatch loop as a generator: with data-parallel replicas it YIELDS the flat gradient of each gradient
        step (already scaled by 1 / world) at the point where it has to be summed over the ranks, and continues with the
        Adam step once the caller has reduced it in place.  `drive_steps` does that with one all-reduce per step;
        `parallel.learn_lockstep` advances several policy groups together and packs their gradients into ONE buffer per
        step.  Returns (StopIteration.value) the training statistics."""
        ids = self._sample_ids(pb, agent)
        n = ids.numel() if ids is not None else pb["obs"].shape[0]
        dev = self.device
        bounds = split_bounds(n, batch_size or -1, merge_last=True)
        mb_start = torch.as_tensor([b[0] for b in bounds] + [n], dtype=torch.int64, device=dev)
        P, A, H = self.net.flat.data, self.net.n_act, self.net.hidden
        n_steps = repeat * len(bounds)
        if self._grad_sync is not None:
            self._grad_sync.require_equal(n_steps, "the number of gradient steps per update")
        scal = torch.zeros(n_steps, 4, dtype=torch.float32, device=dev)
        # the grid is not monotone in the minibatch size (ops.ppo_update_grid): size the slabs for the largest grid
        n_blk_max = max(ops.ppo_update_grid(e - s) for s, e in bounds)
        slabs = self._ws.get(("slabs", n_blk_max))
        if slabs is None:
            slabs = torch.empty(n_blk_max, P.numel(), dtype=torch.float32, device=dev)
            self._ws[("slabs", n_blk_max)] = slabs
        partial = torch.empty(n_blk_max * 4, dtype=torch.float64, device=dev)
        k = 0
        for step in range(repeat):
            if self.recompute_adv and step > 0:  # ppo.py:174-178: returns / advantages only, logp_old stays
                pb = dict(self._preprocess_batch(buffer), logp_old=pb["logp_old"])
            if self.shuffle == "numpy":  # Batch.split draws np.random.permutation (batch.py:1219)
                perm_local = torch.as_tensor(np.random.permutation(n)).to(dev)
            else:
                perm_local = self._device_perm(n, perm_base, agent, repeat, step)
            perm = perm_local if ids is None else ids[perm_local]
            stats = (ops.ppo_adv_stats(pb["adv"], mb_start, perm=perm, max_rows=max(e - s for s, e in bounds))
                     if self.advantage_normalization else None)
            self._global_adv_stats(stats, mb_start)
            for j, (s, e) in enumerate(bounds):
                M = e - s
                nb = ops.ppo_update_grid(M)
                ops.ppo_update_fused(P, pb["obs"], pb["act"], pb["logp_old"], pb["adv"], pb["ret"], self._cfg, A, H,
                                     adv_stats=None if stats is None else stats[j],
                                     v_s_old=pb["v_s"] if self.value_clip else None, perm=perm[s:e], image=self.net.image, M=M,
                                     n_blocks=nb, slabs=slabs[:nb], partial=partial, scalars=scal[k])
                self.opt_step += 1
                grads = slabs[:nb]
                if self._grad_sync is not None and self._grad_sync.fused_step_ok(self.max_grad_norm, P.numel()):
                    # peer-memory path: slab sum, sum over the replicas and Adam in ONE launch (csrc/p2p.hip)
                    self._grad_sync.p2p.adam_step(P, grads, self.exp_avg, self.exp_avg_sq, self.opt_step, lr=self.lr,
                                                  lr_dev=self._lr_dev, betas=self.betas, eps=self.adam_eps,
                                                  weight_decay=self.weight_decay, image=self.net.image, image_map=self.net.image_map)
                    k += 1
                    continue
                if self._grad_sync is not None:  # env-sharded data parallel: ONE flat all-reduce (parallel.py)
                    flat_g = self._ws.setdefault("flat_grad", torch.empty_like(P))
                    ops.reduce_slabs(grads, out=flat_g, scale=1.0 / self._grad_sync.world)  # mean = sum of g_i / world
                    yield flat_g  # summed over the ranks by the driver, in place
                    grads = flat_g.view(1, -1)
                ops.adam_step(P, grads, self.exp_avg, self.exp_avg_sq, self.opt_step, lr=self.lr, lr_dev=self._lr_dev,
                              betas=self.betas, eps=self.adam_eps, weight_decay=self.weight_decay,
                              max_grad_norm=self.max_grad_norm, work=self._adam_work, image=self.net.image,
                              image_map=self.net.image_map)
                k += 1
        self.param_version += 1
        if self._grad_sync is not None:
            self._grad_sync.post_check()  # (peer-memory path: its error word rides behind the update, read below)
        s_h = scal.cpu().numpy()  # the only host sync of the update (the reference does 4 .item() per minibatch)
        if self._grad_sync is not None:
            self._grad_sync.raise_if_failed()
        return A2CTrainingStats(
            loss=SequenceSummaryStats.from_sequence(s_h[:, 0]), actor_loss=SequenceSummaryStats.from_sequence(s_h[:, 1]),
            vf_loss=SequenceSummaryStats.from_sequence(s_h[:, 2]), ent_loss=SequenceSummaryStats.from_sequence(s_h[:, 3]),
            gradient_steps=n_steps)

    # ---- hipGraph path: the whole update (critic passes, GAE, every gradient step) is ONE graph launch ----
    def _warm_kernels(self, buffer: DeviceVectorReplayBuffer) -> None:
        """Run each kernel once on scratch data so that one-time function attributes are set before capture."""
        if self._ws.get("warm"):
            return
        dev, D, A, H = self.device, self.net.obs_dim, self.net.n_act, self.net.hidden
        ops.ensure_scan_workspace(dev)  # (learn() on one long lane: the parallel scan's workspace must exist before any capture)
        P = self.net.flat.data.clone()
        obs = torch.zeros(32, D, device=dev)
        act = torch.zeros(32, dtype=torch.int32, device=dev)
        z = torch.zeros(32, device=dev)
        ops.policy_forward(P, obs, A, H, mode="given", act=act)
        st = ops.ppo_adv_stats(z + torch.arange(32, device=dev), torch.tensor([0, 32], device=dev))
        slabs, _ = ops.ppo_update_fused(P, obs, act, z, z, z, self._cfg, A, H, adv_stats=st[0], v_s_old=z)
        ops.adam_step(P, slabs, torch.zeros_like(P), torch.zeros_like(P), 1, max_grad_norm=self.max_grad_norm,
                      work=self._adam_work)
        ops.gae_lanes(z.view(32, 1), z.view(32, 1), z.view(32, 1), act.view(32, 1).to(torch.uint8),
                      act.view(32, 1).to(torch.uint8))
        torch.cuda.synchronize()
        self._ws["warm"] = True

    def _update_graph(self, buffer: DeviceVectorReplayBuffer, batch_size: int | None, repeat: int):
        T = buffer.host_uniform_len()  # host mirror of the fill level (no device round trip)
        if T is None and self._grad_sync is not None:
            # data parallel: graph or eager must not depend on what the episodes happened to do on THIS rank (the two
            # paths issue different collective sequences) -- only host-known uniform fills (n_step collection) replay
            return None
        if T is None:
            lens_h = buffer.index.lengths.cpu().numpy()
            ins_h = buffer.index.insertion_idx.cpu().numpy()
            T = int(lens_h[0])
            if not (lens_h == T).all() or not (((ins_h - lens_h) % buffer.sub_size) == 0).all():
                return None  # ragged / rotated sub-buffers: eager path with explicit index lists
        if T == 0:
            return None
        no_next = buffer.obs_next_store is None  # ignore_obs_next: V(obs_next) = V(obs) at next(index), see _next_values_by_index
        B, N, D = buffer.buffer_num, buffer.n_agent, buffer.obs_dim
        L, dev = B * N, self.device
        per_agent = self.dispatch == "per_agent"
        groups = list(range(N)) if per_agent else [None]
        n_g = T * B if per_agent else T * L
        bounds = split_bounds(n_g, batch_size or -1, merge_last=True)
        # (the rollout's stored V(obs_next) is that of the TRUE next observation: not what an ignore_obs_next buffer hands out)
        stored = buffer.vnext_store is not None and buffer.policy_outputs_version == self.param_version and not no_next
        key = ("graph", buffer.storage_key(), T, batch_size, repeat, self.dispatch, self.max_grad_norm, stored,
               self._grad_sync is not None, self.graph_collectives, ops.kernel_options())
        g = self._ws.get(key)
        P, A, H = self.net.flat.data, self.net.n_act, self.net.hidden
        if g is None:
            self._warm_kernels(buffer)
            n_steps = len(groups) * repeat * len(bounds)
            if self._grad_sync is not None:
                self._grad_sync.require_equal(n_steps, "the number of gradient steps per update")
            # the grid is not monotone in the minibatch size: size the slabs / partials for the largest grid
            nb_max = max(ops.ppo_update_grid(e - s) for s, e in bounds)
            f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)  # noqa: E731
            w = dict(perm=torch.zeros(len(groups), repeat, n_g, dtype=torch.int64, device=dev),
                     mb_start=torch.as_tensor([b[0] for b in bounds] + [n_g], dtype=torch.int64, device=dev),
                     stats=f(len(groups), repeat, len(bounds), 2), scal=f(n_steps, 4), slabs=f(nb_max, P.numel()),
                     partial=torch.zeros(n_steps, nb_max * 4, dtype=torch.float64, device=dev), nb_max=nb_max,
                     nb_dev=torch.as_tensor([ops.ppo_update_grid(e - s) for s, e in bounds] * (len(groups) * repeat),
                                            dtype=torch.int32, device=dev),
                     M_dev=torch.as_tensor([e - s for s, e in bounds] * (len(groups) * repeat), dtype=torch.int64,
                                           device=dev),
                     perm_ctr=self._perm_ctr,
                     step_dev=torch.zeros(1, dtype=torch.int64, device=dev), v_s=f(T, L), v_next=f(T, L),
                     logp=f(T * L), ret=f(T, L), adv=f(T, L), n_steps=n_steps, flat_g=f(P.numel()))
            obs = buffer.obs_store[:T].reshape(T * L, D)
            obs_next = None if no_next else buffer.obs_next_store[:T].reshape(T * L, D)
            act = buffer.act_store[:T].reshape(T * L)
            rew, term, trunc = (x[:T].reshape(T, L) for x in (buffer.rew_store, buffer.term_store, buffer.trunc_store))

            if stored:
                w["logp"] = buffer.logp_store[:T].reshape(-1)
                if not self.recompute_adv:
                    w["v_s"], w["v_next"] = buffer.vs_store[:T].reshape(T, L), buffer.vnext_store[:T].reshape(T, L)
            if self.advantage_normalization and not self.recompute_adv:
                # minibatch k of (group gi, repeat r) covers perm[gi, r][bounds[k]]: one statistics launch for all of them
                seg = torch.arange(len(groups) * repeat, dtype=torch.int64, device=dev).view(-1, 1) * n_g
                w["mb_start_all"] = torch.cat([(seg + w["mb_start"][:-1].view(1, -1)).reshape(-1),
                                               torch.tensor([len(groups) * repeat * n_g], dtype=torch.int64, device=dev)])

            def preprocess(recompute: bool = False):
                if stored and not recompute:
                    # logp_old / v_s / V(obs_next) were produced by the rollout kernel with these very parameters
                    # (bit-identical to recomputing them as a2c.py:121-127 / ppo.py:157-161 do)
                    # the buffer stores are static allocations: the graph reads them in place (w[...] alias them)
                    if self.recompute_adv:  # later repeats overwrite v_s / v_next: work on copies
                        w["v_s"].copy_(buffer.vs_store[:T].reshape(T, L))
                        w["v_next"].copy_(buffer.vnext_store[:T].reshape(T, L))
                else:
                    # recompute_advantage refreshes the critic values only; logp_old stays (ppo.py:174-178)
                    # before the first Adam step of this update the image may be stale: read `flat` (image=None)
                    img = self.net.image if recompute else None
                    ops.policy_forward(P, obs, A, H, mode="none" if recompute else "given", act=act,
                                       image=img,
                                       out=dict(value=w["v_s"].view(-1), logp=None if recompute else w["logp"],
                                                logits=None))
                    if no_next:
                        self._next_values_by_index(buffer, w["v_s"], T, None, out=w["v_next"])
                    else:
                        ops.policy_forward(P, obs_next, A, H, mode="none", image=img,
                                           out=dict(value=w["v_next"].view(-1), logits=None))
                self._gae(w["v_s"], w["v_next"], rew, term, trunc, N, out=(w["ret"], w["adv"]))

            def body():
                # a generator: everything is launched in order; it yields the tensors that must be summed over the ranks at
                # that point (data parallel only), so that one definition serves the single captured graph (collectives
                # inside), and the segmented form (a graph per stretch between two collectives, collectives eager)
                if self.shuffle == "device":
                    # every permutation of this update (one per agent group and repeat, batch.py:1219) in ONE launch;
                    # the draw counter is the device-resident optimizer step count (it advances by >= one per permutation
                    # and update), so each replay of the graph draws fresh permutations without a counter kernel
                    # (round 4: as a PARALLEL branch of the graph -- a side stream beside preprocess(), it depends on the step
                    #  count alone -- the job got slower, 0.558 -> 0.583 ms per step: a two-stream graph pays more at launch
                    #  than the 6 us kernel it takes off the chain)
                    ops.random_permutations(n_g, len(groups) * repeat, self.seed ^ 0x5DEECE66D, counter_dev=w["step_dev"],
                                            scale=N if per_agent else 1, group_size=repeat,
                                            offset_mul=1 if per_agent else 0, out=w["perm"])
                preprocess()
                if "mb_start_all" in w:
                    ops.ppo_adv_stats(w["adv"], w["mb_start_all"], perm=w["perm"].view(-1), out=w["stats"].view(-1, 2),
                                      max_rows=max(e - s for s, e in bounds))
                    yield from self._global_adv_stats_steps(w["stats"].view(-1, 2), w["mb_start_all"])
                k = 0
                for gi in range(len(groups)):
                    for r in range(repeat):
                        if self.recompute_adv and r > 0:
                            preprocess(recompute=True)
                        perm = w["perm"][gi, r]
                        if self.advantage_normalization and "mb_start_all" not in w:
                            ops.ppo_adv_stats(w["adv"], w["mb_start"], perm=perm, out=w["stats"][gi, r],
                                              max_rows=max(e - s for s, e in bounds))
                            yield from self._global_adv_stats_steps(w["stats"][gi, r], w["mb_start"])
                        for j, (s, e) in enumerate(bounds):
                            nb = ops.ppo_update_grid(e - s)
                            ops.ppo_update_fused(P, obs, act, w["logp"], w["adv"].view(-1), w["ret"].view(-1), self._cfg,
                                                 A, H, adv_stats=w["stats"][gi, r, j] if self.advantage_normalization else None,
                                                 v_s_old=w["v_s"].view(-1) if self.value_clip else None, perm=perm[s:e],
                                                 image=self.net.image if k > 0 else None,  # k == 0: image may be stale
                                                 M=e - s, n_blocks=nb, slabs=w["slabs"][:nb], partial=w["partial"][k],
                                                 want_scalars=False, opt_step_dev=w["step_dev"])
                            grads = w["slabs"][:nb]
                            if self._grad_sync is not None and self._grad_sync.fused_step_ok(self.max_grad_norm, P.numel()):
                                # peer-memory path: slab sum + sum over the replicas + Adam, ONE launch per gradient step
                                self._grad_sync.p2p.adam_step(P, grads, self.exp_avg, self.exp_avg_sq, 1, lr=self.lr,
                                                              lr_dev=self._lr_dev, betas=self.betas, eps=self.adam_eps,
                                                              weight_decay=self.weight_decay, step_dev=w["step_dev"],
                                                              image=self.net.image, image_map=self.net.image_map)
                                k += 1
                                continue
                            if self._grad_sync is not None:  # env-sharded replicas: one captured RCCL all-reduce
                                ops.reduce_slabs(grads, out=w["flat_g"], scale=1.0 / self._grad_sync.world)
                                yield w["flat_g"]  # summed over the ranks, in place
                                grads = w["flat_g"].view(1, -1)
                            ops.adam_step(P, grads, self.exp_avg, self.exp_avg_sq, 1, lr=self.lr, lr_dev=self._lr_dev,
                                          betas=self.betas, eps=self.adam_eps, weight_decay=self.weight_decay,
                                          max_grad_norm=self.max_grad_norm, work=self._adam_work,
                                          step_dev=w["step_dev"], image=self.net.image, image_map=self.net.image_map)
                            k += 1
                # (the 4 loss statistics of every gradient step are folded from w["partial"] by ONE launch after the
                # replay, straight into the pinned slot the host will read)

            def run_inline():  # collectives inside the capture (RCCL) / no collectives at all
                for t_ in body():
                    self._grad_sync.all_reduce_sum_(t_)

            graph = torch.cuda.CUDAGraph()
            if self._grad_sync is not None and not self.graph_collectives:
                # SEGMENTED form: the backend's collectives cannot be captured (gloo), or the capture probe failed.  Every
                # stretch between two collectives is its own hipGraph (one shared memory pool, replayed in capture order);
                # the collectives run eagerly in between: ~20 graph launches + ~19 collectives per update instead of ~80
                # eager kernel launches (one rank over RCCL: 2.5 -> ~1 ms per step).
                pool = torch.cuda.graph_pool_handle()
                segs, gen, more = [], body(), True
                while more:
                    g_ = torch.cuda.CUDAGraph()
                    t_ = None
                    with ops.graph_capture(g_, pool=pool):
                        try:
                            t_ = next(gen)
                        except StopIteration:
                            more = False
                    segs.append((g_, t_))
                w["segments"] = segs
                graph = None
            else:
                self._capture_graph(graph, run_inline)
            w["graph"] = graph
            if self.shuffle == "numpy":
                base = ref_order_rows(T, B, dev)
                w["ref_ids"] = [base * N + a if a is not None else
                                (base[:, None] * N + torch.arange(N, device=dev)[None, :]).reshape(-1) for a in groups]
            self._ws[key] = g = w
        # fresh permutations for this update (Batch.split draws one per repeat, batch.py:1219)
        if self.shuffle == "numpy":
            for gi, a in enumerate(groups):
                for r in range(repeat):
                    pl = torch.as_tensor(np.random.permutation(n_g)).to(dev)
                    # positions of the reference batch (sample(0) order) -> lane ids of the time-major stores
                    g["perm"][gi, r].copy_(g["ref_ids"][gi][pl])
        # shuffle == "device": the permutations are drawn inside the graph (tsm_random_permutations)
        if g.get("step_host") != self.opt_step:  # the device-side step count is stale (eager updates, a loaded checkpoint)
            g["step_dev"].fill_(self.opt_step)
        if g.get("segments") is not None:
            for g_, t_ in g["segments"]:
                g_.replay()
                if t_ is not None:
                    self._grad_sync.all_reduce_sum_(t_)
        else:
            g["graph"].replay()
        self.opt_step += g["n_steps"]
        g["step_host"] = self.opt_step
        self.param_version += 1
        # loss statistics (reference: 4 .item() per minibatch): one launch folds the loss partials of EVERY gradient step
        # and writes the result straight into a pinned (mapped) host slot -- no D2H copy on the stream; the host only
        # blocks when the stats are read
        ring = g.setdefault("ring", [])
        if len(ring) < 4:
            ring.append(dict(h=torch.empty(g["scal"].shape, dtype=torch.float32, pin_memory=True),
                             event=torch.cuda.Event(), pending=None))
            slot = ring[-1]
        else:
            slot = ring[g.get("ring_pos", 0) % 4]
            g["ring_pos"] = g.get("ring_pos", 0) + 1
            if slot["pending"] is not None:
                if self.async_stats:
                    slot["pending"].expire("training stats were not read within 4 update() calls (async_stats=True)")
                else:
                    slot["pending"].resolve()
            # never queue more than 4 updates ahead of the device: an unbounded run-ahead fills the HIP command queue,
            # and the runtime then drains it with a ~2 ms stall every ~10 steps (tools/step_jitter.py)
            slot["event"].synchronize()
        ops.ppo_finalize_many(g["partial"], g["nb_max"] * 4, g["nb_dev"], g["M_dev"], self._cfg, slot["h"])
        sync = self._grad_sync
        if sync is not None:
            sync.post_check()  # a lost peer (peer-memory all-reduce) is reported where the statistics are read: every rank raises
        slot["event"].record()

        def build():
            slot["event"].synchronize()
            if sync is not None:
                sync.raise_if_failed()
            s_h = slot["h"].numpy().copy()
            slot["pending"] = None
            mk = lambda x: A2CTrainingStats(  # noqa: E731
                loss=SequenceSummaryStats.from_sequence(x[:, 0]), actor_loss=SequenceSummaryStats.from_sequence(x[:, 1]),
                vf_loss=SequenceSummaryStats.from_sequence(x[:, 2]), ent_loss=SequenceSummaryStats.from_sequence(x[:, 3]),
                gradient_steps=len(x))
            if per_agent:
                per = len(s_h) // N
                return MapTrainingStats({f"agent_{a}": mk(s_h[a * per:(a + 1) * per]) for a in range(N)})
            return mk(s_h)

        out = LazyStats(build)
        slot["pending"] = out
        if not self.async_stats:
            out.resolve()
        return out

    def update(self, buffer: DeviceVectorReplayBuffer, batch_size: int | None, repeat: int):
        """OnPolicyAlgorithm.update -> Algorithm._update (algorithm_base.py:852-863, 584-629): raises outside a training
        step (:610-615); steps every LR scheduler once per update (:626-627)."""
        if not self.is_within_training_step:
            raise RuntimeError(
                "update() was called outside of a training step as signalled by `is_within_training_step=False`; "
                "wrap the call in `policy_within_training_step(policy)` (tianshou/utils/torch_utils.py:31-46)")
        t0 = time.time()
        out = self._update(buffer, batch_size, repeat, t0)
        for sched in self.lr_schedulers:
            sched.step()
        out.train_time = time.time() - t0
        return out

    def _update(self, buffer: DeviceVectorReplayBuffer, batch_size: int | None, repeat: int, t0: float):
        # `flat` is the source of truth and may have been written from outside (load_state_dict, broadcast, tests); the
        # padded image is a cache that the Adam kernel refreshes.  The graph path reads `flat` itself until its first
        # Adam step has rewritten the image, so it needs no refresh launch; the eager path refreshes it here.
        # data parallel without capturable collectives: segmented graphs (TSM_SEGMENTED=0: plain eager launches)
        if self.use_graph and (self._grad_sync is None or self.graph_collectives or os.environ.get("TSM_SEGMENTED", "1") != "0"):
            out = self._update_graph(buffer, batch_size, repeat)
            if out is not None:
                return out
        self.net.sync_image()
        pb = self._preprocess_batch(buffer)
        # device-side shuffling: the draws are numbered from the optimizer step count at the start of the update, as in
        # the captured update (`_device_perm`); ragged buffers list their rows explicitly and keep the same numbering
        perm_base = self.opt_step
        if self.dispatch == "per_agent":
            # MARLDispatcher.dispatch_update_with_batch: the (shared) algorithm is updated once per agent id,
            # each time on that agent's rows only (marl.py:251-268); stats keyed "{agent}/..." (marl.py:51-59)
            per_agent = {}
            for a in range(buffer.n_agent):
                st = self._update_with_batch(pb, batch_size, repeat, agent=a, buffer=buffer, perm_base=perm_base)
                st.train_time = time.time() - t0
                per_agent[f"agent_{a}"] = st
            return MapTrainingStats(per_agent)
        return self._update_with_batch(pb, batch_size, repeat, agent=None, buffer=buffer, perm_base=perm_base)

    # ---- `.learn(batch)` for the MARL trainers (training_coordinator.py:336) ----------------------
    def learn_steps(self, batch: Batch, batch_size: int | None = None, repeat: int = 1, **kwargs):
        """`learn` as a generator of gradient synchronisation points (see `_update_steps`)."""
        dev = self.device
        t = lambda x, dt: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))).to(dev, dt).contiguous()  # noqa: E731
        self.net.sync_image()
        if self._batch_store(batch) is not None:  # rows read in place by the graph path: the eager path takes the copies
            batch = self._batch_store(batch).agent_batch(int(batch.agent_index))
        obs = t(batch.obs, torch.float32)
        n = obs.shape[0]
        act = t(batch.act, torch.int32).reshape(n)
        P = self.net.flat.data
        if self._stored_outputs_ok(batch):
            # the rollout kernel stored these rows' logp / V(obs) / V(obs_next), computed by this policy at this parameter version:
            # the same bits as the three passes below (tests/test_gpu_tag.py), without them
            cur = dict(logp=t(batch.logp_old, torch.float32).reshape(n), value=t(batch.v_s, torch.float32).reshape(n))
            nxt = dict(value=t(batch.v_next, torch.float32).reshape(n))
        else:
            cur = ops.policy_forward(P, obs, self.net.n_act, self.net.hidden, image=self.net.image, mode="given", act=act, want_logits=False)
            nxt = ops.policy_forward(P, t(batch.obs_next, torch.float32), self.net.n_act, self.net.hidden, image=self.net.image, mode="none",
                                     want_logits=False)
        term = t(batch.terminated, torch.uint8).reshape(n, 1)
        trunc = t(batch.truncated, torch.uint8).reshape(n, 1) if "truncated" in batch else torch.zeros_like(term)
        ret, adv = ops.gae_lanes(cur["value"].view(n, 1), nxt["value"].view(n, 1), t(batch.rew, torch.float32).view(n, 1),
                                 term, trunc, self.gamma, self.gae_lambda)
        pb = dict(T=n, rows=None, obs=obs, act=act, v_s=cur["value"], ret=ret.reshape(-1), adv=adv.reshape(-1),
                  logp_old=cur["logp"], n_env=1, n_agent=1)
        st = yield from self._update_steps(pb, batch_size, repeat)
        return {"loss": st.loss.mean, "actor_loss": st.actor_loss.mean, "vf_loss": st.vf_loss.mean,
                "ent_loss": st.ent_loss.mean}

    def learn(self, batch: Batch, batch_size: int | None = None, repeat: int = 1, **kwargs) -> dict[str, float]:
        """One PPO pass on an explicit agent batch holding obs, act, rew, obs_next, terminated[, truncated].
        Rows are one time-ordered lane (the reference's per-agent Batch); GAE treats the last row as end.
        Single GPU: ONE hipGraph replay per call.  Data-parallel replicas: the same static-buffer body with its collectives
        captured (RCCL) or between segmented graphs (`parallel.learn_lockstep_graph` with this one policy)."""
        if self.learn_graph_ok(repeat):
            if self._grad_sync is None:
                return self._learn_graph(batch, batch_size, repeat)
            from ..parallel import learn_lockstep_graph, lockstep_graphs_enabled

            if lockstep_graphs_enabled():
                return learn_lockstep_graph([(self, batch, batch_size, repeat)], self._grad_sync)[0]
        return drive_steps(self.learn_steps(batch, batch_size, repeat, **kwargs), self._grad_sync)

    @staticmethod
    def _batch_store(batch: Batch):
        """The `DeviceStoreRows` behind an agent batch built with `agent_batches_from_buffer(copies=False, global_state=False)`
        (the rows are read from the device buffer's stores in place), else None."""
        return getattr(batch["store_rows"], "store", None) if "store_rows" in batch else None

    def _learn_rows(self, batch: Batch) -> int:
        st = self._batch_store(batch)
        return st.T * st.E if st is not None else len(batch.rew)

    def _stored_outputs_ok(self, batch: Batch) -> bool:
        """Does `batch` (training_coordinator.agent_batches_from_buffer) carry logp_old / v_s / v_next that THIS policy computed at
        its CURRENT parameters?  (A shared policy that learns once per agent has moved on after its first call: the later calls
        recompute, as the reference does, ppo.py:157-161.)"""
        if not self.reuse_rollout_outputs:
            return False
        st = self._batch_store(batch)
        if st is not None:
            cols = st.column_outputs
            return bool(cols is not None and st.vnext is not None and st.logp is not None
                        and cols[int(batch.agent_index)] == (id(self), self.param_version))
        return bool("v_next" in batch and "logp_old" in batch and "v_s" in batch
                    and "outputs_policy" in batch and int(batch.outputs_policy) == id(self)
                    and int(batch.outputs_version) == self.param_version)

    def learn_graph_ok(self, repeat: int = 1) -> bool:
        """Can `learn` run from static buffers inside captured graphs?  (recompute_advantage re-runs the critic between
        repeats from the host.)"""
        return bool(self.use_graph and not (self.recompute_adv and repeat > 1))

    def _learn_static(self, n: int, batch_size: int | None, repeat: int, has_trunc: bool, stored: bool = False) -> dict:
        """Static HBM buffers + the launch sequence of one `learn` call on n rows (cached per shape): `w["body"]()` is a
        generator that issues the critic passes, GAE, the permutations, advantage statistics and every gradient step on
        the static buffers -- the same launches in the same order as `learn_steps`, hence the same bits -- and, for a
        data-parallel replica, YIELDS every tensor that has to be summed over the ranks (the advantage-statistics pack,
        the flat gradient of each step) exactly where `learn_steps` does.  The optimizer step count and the permutation
        counter live in HBM, so replays advance them; the learning rate is read from HBM."""
        dev = self.device
        D, A, H = self.net.obs_dim, self.net.n_act, self.net.hidden
        dp = self._grad_sync is not None
        key = ("learn_graph", n, batch_size, repeat, has_trunc, self.shuffle, dp, ops.kernel_options(), stored)
        w = self._ws.get(key)
        if w is not None:
            return w
        bounds = split_bounds(n, batch_size or -1, merge_last=True)
        n_steps = repeat * len(bounds)
        self._warm_kernels(None)
        P = self.net.flat.data
        z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device=dev)  # noqa: E731
        nb_max = max(ops.ppo_update_grid(e - s) for s, e in bounds)
        w = dict(n=n, n_steps=n_steps, repeat=repeat, has_trunc=has_trunc, stored=stored,
                 obs=z(n, D), obs_next=None if stored else z(n, D), act=z(n, dt=torch.int32), rew=z(n, 1), term=z(n, 1, dt=torch.uint8),
                 trunc=z(n, 1, dt=torch.uint8), step_dev=z(1, dt=torch.int64),
                 slabs=torch.empty(nb_max, P.numel(), dtype=torch.float32, device=dev),
                 # the loss partials of EVERY gradient step stay until ONE launch behind the replay folds them, straight into the
                 # pinned slot the host reads (as update() does): no per-step finalize launch, no device -> host copy
                 partial=torch.zeros(n_steps, nb_max * 4, dtype=torch.float64, device=dev), nb_max=nb_max,
                 nb_dev=torch.as_tensor([ops.ppo_update_grid(e - s) for s, e in bounds] * repeat, dtype=torch.int32, device=dev),
                 M_dev=torch.as_tensor([e - s for s, e in bounds] * repeat, dtype=torch.int64, device=dev),
                 perm=z(repeat, n, dt=torch.int64), perm_done=z(1, dt=torch.int32),
                 mb_start=torch.as_tensor([b[0] for b in bounds] + [n], dtype=torch.int64, device=dev))
        if stored:  # logp_old / V(obs) / V(obs_next) as the rollout stored them (see _stored_outputs_ok)
            w.update(logp=z(n), v_s=z(n), v_next=z(n))
        if dp:
            w["flat_g"] = z(P.numel())

        def body():
            if stored:
                cur, nxt = dict(logp=w["logp"], value=w["v_s"]), dict(value=w["v_next"])
            else:
                # `flat` is the source of truth before the first Adam step of this call (the image may be stale)
                cur = ops.policy_forward(P, w["obs"], A, H, image=None, mode="given", act=w["act"], want_logits=False)
                nxt = ops.policy_forward(P, w["obs_next"], A, H, image=None, mode="none", want_logits=False)
            ret, adv = ops.gae_lanes(cur["value"].view(n, 1), nxt["value"].view(n, 1), w["rew"], w["term"], w["trunc"],
                                     self.gamma, self.gae_lambda)
            ret, adv = ret.reshape(-1), adv.reshape(-1)
            k = 0
            for r in range(repeat):
                if self.shuffle != "numpy":
                    # (the launch advances the draw counter itself: the draws of `_device_perm`, one launch fewer)
                    ops.random_permutations(n, 1, self.seed ^ 0x5DEECE66D, counter_dev=self._perm_ctr, out=w["perm"][r:r + 1],
                                            advance=1, done_ctr=w["perm_done"])
                perm = w["perm"][r]
                stats = (ops.ppo_adv_stats(adv, w["mb_start"], perm=perm, max_rows=max(e - s for s, e in bounds))
                         if self.advantage_normalization else None)
                if dp:  # the union minibatch's statistics: ONE f64 pack summed over the ranks
                    yield from self._global_adv_stats_steps(stats, w["mb_start"])
                for j, (s_, e_) in enumerate(bounds):
                    nb = ops.ppo_update_grid(e_ - s_)
                    ops.ppo_update_fused(P, w["obs"], w["act"], cur["logp"], adv, ret, self._cfg, A, H,
                                         adv_stats=None if stats is None else stats[j],
                                         v_s_old=cur["value"] if self.value_clip else None, perm=perm[s_:e_],
                                         image=self.net.image if k > 0 else None, M=e_ - s_, n_blocks=nb,
                                         slabs=w["slabs"][:nb], partial=w["partial"][k], want_scalars=False,
                                         opt_step_dev=w["step_dev"])
                    grads = w["slabs"][:nb]
                    if dp and self._grad_sync.fused_step_ok(self.max_grad_norm, P.numel()):
                        self._grad_sync.p2p.adam_step(P, grads, self.exp_avg, self.exp_avg_sq, 1, lr=self.lr, lr_dev=self._lr_dev,
                                                      betas=self.betas, eps=self.adam_eps, weight_decay=self.weight_decay,
                                                      step_dev=w["step_dev"], image=self.net.image, image_map=self.net.image_map)
                        k += 1
                        continue
                    if dp:
                        ops.reduce_slabs(grads, out=w["flat_g"], scale=1.0 / self._grad_sync.world)
                        yield w["flat_g"]  # summed over the ranks, in place
                        grads = w["flat_g"].view(1, -1)
                    ops.adam_step(P, grads, self.exp_avg, self.exp_avg_sq, 1, lr=self.lr, lr_dev=self._lr_dev,
                                  betas=self.betas, eps=self.adam_eps, weight_decay=self.weight_decay,
                                  max_grad_norm=self.max_grad_norm, work=self._adam_work, step_dev=w["step_dev"],
                                  image=self.net.image, image_map=self.net.image_map)
                    k += 1

        w["body"] = body
        self._ws[key] = w
        return w

    def _learn_load(self, w: dict, batch: Batch) -> None:
        """The batch into the static buffers (copy_ converts int64 actions / bool flags; host arrays are uploaded), this
        call's host-drawn permutations, and the device-side step count if it went stale."""
        n, D, repeat = w["n"], self.net.obs_dim, w["repeat"]
        t = lambda x: x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))  # noqa: E731
        stored = w.get("stored", False)
        st = self._batch_store(batch)
        if st is not None:
            # the agent's column straight from the time-major stores into the static buffers: ONE launch whose descriptors are
            # built once per (store, column) -- every pointer in them is a static allocation
            a = int(batch.agent_index)
            cache = w.setdefault("store_gathers", {})
            prep = cache.get((st.key, a))
            if prep is None:
                T_, E_, N_, D_ = st.T, st.E, st.N, st.D
                fields = [(st.obs, w["obs"], T_, E_, N_ * D_, a * D_), (st.act, w["act"], T_, E_, N_, a), (st.rew, w["rew"], T_, E_, N_, a),
                          (st.term, w["term"], T_, E_, N_, a), (st.trunc, w["trunc"], T_, E_, N_, a)]
                fields += ([(st.logp, w["logp"], T_, E_, N_, a), (st.vs, w["v_s"], T_, E_, N_, a), (st.vnext, w["v_next"], T_, E_, N_, a)]
                           if stored else [(st.obs_next, w["obs_next"], T_, E_, N_ * D_, a * D_)])
                cache[(st.key, a)] = ops.gather_fields(fields)
            else:
                ops.gather_fields(None, prepared=prep)
            self._learn_load_rest(w, n, repeat)
            return
        names = ["obs", "act", "rew", "terminated"] + (["truncated"] if "truncated" in batch else []) + \
            (["logp_old", "v_s", "v_next"] if stored else ["obs_next"])
        leaves = [batch[k] for k in names]
        if all(isinstance(x, torch.Tensor) and x.is_cuda and x.is_contiguous() and x.dtype in ops._GATHER_KINDS for x in leaves) \
                and all(batch[k].dtype == torch.float32 for k in names if k not in ("act", "terminated", "truncated")):
            # device batches (the trainers' per-agent batches): all fields into the static buffers in ONE launch
            dst = [w["obs"], w["act"], w["rew"], w["term"]] + ([w["trunc"]] if "truncated" in batch else []) + \
                ([w["logp"], w["v_s"], w["v_next"]] if stored else [w["obs_next"]])
            ops.gather_fields(list(zip(leaves, dst)))
            self._learn_load_rest(w, n, repeat)
            return
        w["obs"].copy_(t(batch.obs).reshape(n, D), non_blocking=True)
        if stored:
            for k_, d_ in (("logp_old", "logp"), ("v_s", "v_s"), ("v_next", "v_next")):
                w[d_].copy_(t(batch[k_]).reshape(n), non_blocking=True)
        else:
            w["obs_next"].copy_(t(batch.obs_next).reshape(n, D), non_blocking=True)
        w["act"].copy_(t(batch.act).reshape(n), non_blocking=True)
        w["rew"].copy_(t(batch.rew).reshape(n, 1), non_blocking=True)
        w["term"].copy_(t(batch.terminated).reshape(n, 1), non_blocking=True)
        if "truncated" in batch:
            w["trunc"].copy_(t(batch.truncated).reshape(n, 1), non_blocking=True)
        self._learn_load_rest(w, n, repeat)

    def _learn_load_rest(self, w: dict, n: int, repeat: int) -> None:
        if self.shuffle == "numpy":  # Batch.split draws np.random.permutation per repeat (batch.py:1219)
            for r in range(repeat):
                w["perm"][r].copy_(torch.as_tensor(np.random.permutation(n)), non_blocking=True)
        if w.get("step_host") != self.opt_step:  # the device-side step count is stale (eager updates, a loaded checkpoint)
            w["step_dev"].fill_(self.opt_step)

    def _learn_finish(self, w: dict):
        """Behind the replay: host-side counters, and the statistics on their way to a pinned host slot.
        async_stats=True: the returned mapping waits for them only when it is read (the reference's learn() returns floats:
        4 .item() per minibatch) -- the host never blocks here, except to keep at most 4 calls in flight."""
        n_steps = w["n_steps"]
        self.opt_step += n_steps
        w["step_host"] = self.opt_step
        self.param_version += 1
        ring = w.setdefault("ring", [])
        if len(ring) < 4:
            ring.append(dict(h=torch.empty(n_steps, 4, dtype=torch.float32, pin_memory=True), event=torch.cuda.Event(),
                             pending=None))
            slot = ring[-1]
        else:
            slot = ring[w.get("ring_pos", 0) % 4]
            w["ring_pos"] = w.get("ring_pos", 0) + 1
            if slot["pending"] is not None:
                slot["pending"]._force()
        if "scal" in w:  # (GenericPPO._learn_static folds its statistics inside the graph)
            slot["h"].copy_(w["scal"], non_blocking=True)
        else:
            ops.ppo_finalize_many(w["partial"], w["nb_max"] * 4, w["nb_dev"], w["M_dev"], self._cfg, slot["h"])
        slot["sync"] = self._grad_sync
        if self._grad_sync is not None:
            self._grad_sync.post_check()
        slot["event"].record()
        out = LazyLosses(slot)
        if not self.async_stats:  # plain floats at once, as the reference returns them
            return dict(out)
        slot["pending"] = out
        return out

    def _learn_graph(self, batch: Batch, batch_size: int | None, repeat: int) -> dict[str, float]:
        """`learn` as ONE hipGraph replay per call (the MARL trainers call it once per policy and step,
        training_coordinator.py:118,154,336): the batch is copied into static HBM buffers, then `_learn_static`'s body
        replays as captured."""
        w = self._learn_static(self._learn_rows(batch), batch_size, repeat, "truncated" in batch or self._batch_store(batch) is not None,
                               stored=self._stored_outputs_ok(batch))
        self._learn_load(w, batch)
        if "graph" not in w:
            graph = torch.cuda.CUDAGraph()
            with ops.graph_capture(graph):
                for _ in w["body"]():  # (no replica: nothing is yielded)
                    raise RuntimeError("a single-GPU learn() has no collectives")
            w["graph"] = graph
        w["graph"].replay()
        return self._learn_finish(w)

    def __deepcopy__(self, memo):
        """Snapshot for opponent pools (training_coordinator.py:481-494): parameters, optimizer state and counters are
        copied; device workspaces (captured graphs, pinned rings) are rebuilt lazily by the copy."""
        net = DiscreteActorCritic(self.net.obs_dim, self.net.n_act, self.net.hidden, device=self.device)
        net.flat.data.copy_(self.net.flat.data)
        net.sync_image()
        net._ref_keys = getattr(self.net, "_ref_keys", None)
        new = PPO(net=net, **self._ctor)
        new.load_state_dict(self.state_dict())
        new.train(self.training)
        return new

    # ---- checkpointing: the reference's layout (algorithm_base.py:521-541) --------------------------------------
    def state_dict(self, *args, **kwargs):
        """`Algorithm.state_dict()` of the reference: the nn.Module parameters under their module paths
        (`policy.actor.preprocess.model.model.0.weight` ... `critic.last.model.0.bias`) plus `_optimizers` = a list with
        ONE torch-Adam `state_dict()` (`state[i] = {step, exp_avg, exp_avg_sq}` per parameter in
        ActorCritic.parameters() order, `param_groups`), so checkpoints move between the reference and this engine in both
        directions (pinned by tests/golden/checkpoint.npz).  What the engine needs beyond that to resume bit-identically
        (return statistics, sampling / permutation counters, scheduler epochs) rides in `_optimizers[0]["tsm_engine"]`,
        a key torch's `Optimizer.load_state_dict` ignores."""
        from collections import OrderedDict

        views = self.net.reference_named_views()
        sd = OrderedDict((k, v.detach().clone()) for k, v in views)
        state, o = {}, 0
        for i, (_, v) in enumerate(views):
            n = v.numel()
            if self.opt_step > 0:  # torch creates the per-parameter state at the first step
                state[i] = {"step": torch.tensor(float(self.opt_step)),
                            "exp_avg": self.exp_avg[o:o + n].view(v.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(v.shape).clone()}
            o += n
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.adam_eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False, "params": list(range(len(views)))}
        if self.lr_schedulers:
            group["initial_lr"] = self.lr_schedulers[0].base_lr  # torch's LambdaLR stores it in the group
        engine = {"ret_rms": self.ret_rms.dev.detach().cpu().clone(), "sample_ctr": int(self._sample_ctr),
                  "perm_ctr": int(self._perm_ctr.item()), "opt_step": int(self.opt_step),
                  "lr_schedulers": [s.state_dict() for s in self.lr_schedulers]}
        sd["_optimizers"] = [{"state": state, "param_groups": [group], "tsm_engine": engine}]
        return sd

    def load_state_dict(self, sd, *args, **kwargs):
        views = self.net.reference_named_views()
        missing = [k for k, _ in views if k not in sd]
        if missing:
            raise KeyError(f"state_dict lacks {missing[:3]}{' ...' if len(missing) > 3 else ''}")
        with torch.no_grad():
            for k, v in views:
                v.copy_(torch.as_tensor(sd[k]).to(v.device, v.dtype).reshape(v.shape))
        self.net.sync_image()
        self.param_version += 1
        opt = sd["_optimizers"][0]
        st, o, step = opt["state"], 0, 0
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, (_, v) in enumerate(views):
            n = v.numel()
            e = st.get(i, st.get(str(i)))
            if e is not None:
                self.exp_avg[o:o + n].copy_(torch.as_tensor(e["exp_avg"]).reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(torch.as_tensor(e["exp_avg_sq"]).reshape(-1))
                step = int(float(e["step"]))
            o += n
        self.opt_step = step
        g = opt["param_groups"][0]
        self.lr = float(g["lr"])
        hyper = (tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"]))
        if hyper != (tuple(self.betas), self.adam_eps, self.weight_decay):
            self._ws = {}  # captured graphs hold the old Adam constants as kernel arguments
        self.betas, self.adam_eps, self.weight_decay = hyper
        eng = opt.get("tsm_engine")
        if eng is not None:
            self.ret_rms.dev.copy_(torch.as_tensor(eng["ret_rms"]))
            self._sample_ctr = int(eng["sample_ctr"])
            self._perm_ctr.fill_(int(eng["perm_ctr"]))
            for s, s_sd in zip(self.lr_schedulers, eng.get("lr_schedulers", [])):
                s.load_state_dict(s_sd)

    # ---- the reference's training entry points (algorithm_base.py:543-582) ------------------------------------------
    def create_trainer(self, params):
        from ..trainer import OnPolicyTrainer

        return OnPolicyTrainer(self, params)

    def run_training(self, params):
        return self.create_trainer(params).run()


class LazyLosses(dict):
    """What `learn()` returns with `async_stats=True` ({"loss", "actor_loss", "vf_loss", "ent_loss"}: means over the call's
    gradient steps): a dict filled in when it is first read -- the statistics are on their way to pinned host memory behind
    the captured update.  Every Python-level read resolves it; C code that walks the dict storage directly (json.dumps
    without `indent`) does not, so pass `dict(x)` to such consumers."""

    def __init__(self, slot: dict) -> None:
        super().__init__()
        self._slot = slot

    def _force(self) -> None:
        slot = self._slot
        if slot is None:
            return
        self._slot = None
        slot["event"].synchronize()
        if slot.get("sync") is not None:
            slot["sync"].raise_if_failed()
        if ops.gae_scan_failed():
            raise RuntimeError("GAE: a workgroup of the parallel long-series scan gave up waiting for another one's map "
                               "(returns / advantages of that learn() call are NaN)")
        s_h = slot["h"].numpy()
        dict.update(self, loss=float(s_h[:, 0].mean()), actor_loss=float(s_h[:, 1].mean()), vf_loss=float(s_h[:, 2].mean()),
                    ent_loss=float(s_h[:, 3].mean()))
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

    def __eq__(self, other):
        self._force()
        if isinstance(other, LazyLosses):
            other._force()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        self._force()
        return dict.__repr__(self)

    def copy(self):
        self._force()
        return dict(self)


def drive_steps(gen, grad_sync):
    """Run an `_update_steps` / `learn_steps` generator to its end: every yielded flat gradient is summed over the
    data-parallel ranks in place (one all-reduce per gradient step).  Single-process jobs never yield."""
    try:
        while True:
            flat_g = next(gen)
            grad_sync.all_reduce_sum_(flat_g)
    except StopIteration as stop:
        return stop.value


def ref_order_rows(T: int, B: int, device) -> torch.Tensor:
    """Joint-row ids (slot * B + env) of a uniformly filled buffer listed in the reference's sample(0) order:
    sub-buffer after sub-buffer, time-ordered inside each (manager.py:224-229)."""
    return (torch.arange(T, device=device)[None, :] * B + torch.arange(B, device=device)[:, None]).reshape(-1)


class policy_within_training_step:
    """tianshou.utils.torch_utils.policy_within_training_step (torch_utils.py:31-46)."""

    def __init__(self, policy, enabled: bool = True) -> None:
        self.policy, self.enabled = policy, enabled

    def __enter__(self):
        self.prev = self.policy.is_within_training_step
        self.policy.is_within_training_step = self.enabled
        return self

    def __exit__(self, *exc):
        self.policy.is_within_training_step = self.prev
        return False
