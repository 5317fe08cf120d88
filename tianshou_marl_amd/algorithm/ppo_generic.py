"""GenericPPO -- the same PPO as `algorithm/ppo.py`, for actor / critic MLPs of ANY width (and a centralized critic).

`PPO` runs on the fused 64-wide kernels (csrc/mlp_fused.hip: hidden size 64, obs <= 64).  Reference users configure
`Net(hidden_sizes=[128, 128])`, tanh nets, deeper nets, or a centralized critic over the concatenated global state; for
those this class composes the general kernels instead:
    actor / critic forward + backward   csrc/dense.hip      (f32-MFMA tiled GEMMs, `FlatMLP`)
    Categorical sample / log-prob       csrc/categorical.hip
    GAE                                 csrc/gae.hip
    PPO loss forward + backward         csrc/ppo_loss.hip   (on given logits / value -> dlogits, dvalue)
    clip_grad_norm_ + Adam              csrc/adam.hip       (ONE optimizer over actor + critic, as ActorCritic does)
Interface, hyper-parameters, per-agent dispatch, statistics and the reference call sites are those of `PPO`
(ppo.py:17-224, a2c.py:113-151, marl.py:208-268).  With `graph=True` (the default) the second update of a shape captures
the whole sequence -- permutations, critic passes, GAE, every gradient step -- into ONE hipGraph that later updates replay;
128-wide nets take the one-launch row kernels of csrc/ppo_rows.hip (`fused_actor` / `fused_critic`).
`critic_input="global"`: the critic sees the env's joint observation `[N * obs_dim]` (GlobalStateConstructor
"concatenate", ctde.py:291-294) and its single value is shared by the env's agents (centralized-critic PPO).
"""
from __future__ import annotations

import copy
import os
from typing import Literal

import numpy as np
import torch

from .. import ops
from ..data.batch import Batch, split_bounds
from ..data.buffer import DeviceVectorReplayBuffer
from ..data.stats import A2CTrainingStats, SequenceSummaryStats
from ..utils.net import FlatMLP, MLPActorCritic
from .ppo import PPO, drive_steps, ref_order_rows


class GenericPPO(PPO):

    def __init__(self, *, net: MLPActorCritic | None = None, critic_input: Literal["local", "global"] | None = None,
                 n_agent: int | None = None, graph: bool = True, fused_actor: bool = True,
                 reuse_rollout_outputs: bool = True, shift_next_values: bool = True, **kwargs) -> None:
        if net is None and kwargs.get("policy") is not None:  # reference-style construction (see PPO.__new__)
            from ..utils.net import net_from_reference_modules

            net = net_from_reference_modules(kwargs["policy"], kwargs.get("critic"), kwargs.get("device", "cuda"))
            # what PPO.__init__ takes from the reference objects when it builds the net itself (ppo.py:55-133): greedy
            # evaluation as the policy says, and an update on the whole batch handed over (MARLDispatcher's per-agent
            # sequence is `dispatch="per_agent"`)
            kwargs["deterministic_eval"] = bool(getattr(kwargs["policy"], "deterministic_eval",
                                                        kwargs.get("deterministic_eval", False)))
            kwargs["dispatch"] = kwargs.get("dispatch") or "pooled"
        if not isinstance(net, MLPActorCritic):
            raise TypeError("GenericPPO needs an MLPActorCritic (use PPO for DiscreteActorCritic)")
        if critic_input is None:  # a critic that is wider than the actor's observation sees the env's joint observation
            critic_input = "local" if net.critic_obs_dim == net.obs_dim else "global"
        if n_agent is None:
            n_agent = max(1, net.critic_obs_dim // net.obs_dim) if critic_input == "global" else 1
        kwargs["use_graph"] = False  # the base class's graph path is the fused-kernel one
        super().__init__(net=net, **kwargs)
        self.graph = bool(graph)  # capture this class's own update sequence into one hipGraph
        self.critic_input, self.n_agent = critic_input, int(n_agent)
        want = net.obs_dim * (self.n_agent if critic_input == "global" else 1)
        if net.critic_obs_dim != want:
            raise ValueError(f"critic input width {net.critic_obs_dim} != {want} for critic_input={critic_input!r}")
        self._ctor.update(critic_input=critic_input, n_agent=n_agent, graph=graph)
        self._cfg_rows = ops.make_ppo_cfg(self.eps_clip, self.dual_clip, self.value_clip, self.advantage_normalization,
                                          self.vf_coef, self.ent_coef, value_group=self.n_agent)
        # value term alone (the actor's terms come from the one-launch actor step, csrc/ppo_rows.hip)
        self._cfg_value = {vg: ops.make_ppo_cfg(self.eps_clip, self.dual_clip, self.value_clip, False, self.vf_coef,
                                                self.ent_coef, loss_kind=2, value_group=vg) for vg in (1, self.n_agent)}
        self.fused_actor = bool(fused_actor) and ops.ppo_actor_rows_supported(net.obs_dim, net.actor.dims[1:-1], net.n_act,
                                                                              net.actor.act)
        # ... and the critic in its own single launch when it is in_dim -> 128 -> 128 -> 1 (row minibatches or a local
        # critic: the samples of a row are adjacent lanes)
        n_val = self.n_agent if critic_input == "global" else 1
        gen1_ok = ops.ppo_critic_rows_supported(net.critic_obs_dim, net.critic.dims[1:-1], n_val, net.critic.act)  # (widths <= 96)
        gen2_ok = (1 <= n_val <= 16 and os.environ.get("TSM_CRITIC_GEN", "2") != "1"
                   and ops.critic_rows_grad_supported(net.critic_obs_dim, net.critic.dims[1:-1], 1, net.critic.act))
        self.fused_critic = bool(self.fused_actor and net.critic.dims[-1] == 1 and (gen1_ok or gen2_ok))
        # second generation of the critic step (csrc/critic_train.hip + critic_dw1.hip: two launches, dW1 as a split-K pass)
        self.critic_gen2 = self.fused_critic and gen2_ok
        # ... optionally with dW2 formed by the split-K pass too (round 4 experiment, TSM_CRITIC_SPLIT_DW2=1: 14 MB less HBM
        # traffic per step but 2 us SLOWER -- the latency-bound dW1 launch grows by more than the tile kernel shrinks; DESIGN.md
        # section 4), default: a rank-32 dW2 slab per tile, side-reduced inside the dW1 launch
        self.split_dw2 = self.critic_gen2 and os.environ.get("TSM_CRITIC_SPLIT_DW2", "0") == "1"
        # ... and V(row) of such a critic for all rows of a pass in one launch (csrc/critic_rows.hip) instead of three GEMMs
        self.fused_values = bool(fused_actor) and net.critic.dims[-1] == 1 and ops.critic_rows_forward_supported(
            net.critic_obs_dim, net.critic.dims[1:-1], 1, net.critic.act)
        if self.fused_values:
            ops.call("tsm_critic_rows_init", net.critic_obs_dim, net.critic.dims[1])  # (function attributes: before any capture)
            ops._critic_rows_ready.add((net.critic_obs_dim, net.critic.dims[1]))
        # the critic's first-layer weights in the gradient kernel's fragment order (csrc/critic_train.hip): written by the
        # segmented Adam launch of every step, read by the NEXT step of the same update (`_w1_img_ok`: cleared when an update
        # begins -- its first step gathers from the flat vector, which is the source of truth)
        self._w1_img = ops.critic_w1_image(net.critic.flat.data, net.critic_obs_dim) if self.critic_gen2 else None
        self._w1_img_ok = False
        self.reuse_rollout_outputs = bool(reuse_rollout_outputs)
        # V(obs_next) of chained rows from V(obs) of the next slot instead of a second full critic pass (_next_values_chained)
        self.shift_next_values = bool(shift_next_values)
        self._ctor.update(fused_actor=fused_actor, reuse_rollout_outputs=reuse_rollout_outputs,
                          shift_next_values=shift_next_values)

    # ---- helpers --------------------------------------------------------------------------------------------------
    @property
    def row_minibatches(self) -> bool:
        """Centralized critic + pooled dispatch: a minibatch is a set of JOINT ROWS (one env step = the reference's buffer
        row, all N agents of it) instead of a set of lanes.  The actor then runs on the row's N observations and the
        critic ONCE on their concatenation -- the same 4 * N * D bytes serve both nets, and the critic costs 1 / N of the
        per-lane form.  `batch_size` still counts samples (lanes) and must be a multiple of N."""
        return self.critic_input == "global" and self.dispatch == "pooled"

    def _critic_values(self, x: torch.Tensor, run_if: torch.Tensor | None = None) -> torch.Tensor:
        """critic(x) for rows x [n, critic_obs_dim] -> [n].  One launch with activations on-chip for 128-wide critics
        (`fused_values`), else the dense GEMM chain.  run_if (device flag): the pass runs only if it is non-zero (the result
        is garbage otherwise).  Every value this class compares with another one comes from this one function."""
        critic = self.net.critic
        if self.fused_values:
            return ops.critic_rows_forward(critic.flat.data, x, critic.dims[1], run_if=run_if)
        if run_if is not None:
            return ops.mlp_forward_cond(critic.desc, critic.flat.data, x, run_if)[0].reshape(-1)
        return FlatMLP.forward(critic, x, save=False).reshape(-1)

    def _unit_values(self, obs_rows: torch.Tensor, joint: torch.Tensor | None) -> torch.Tensor:
        """V per critic UNIT: a lane row (local critic) or a joint row of the env step (centralized critic)."""
        return self._critic_values(obs_rows if self.critic_input == "local" else joint)

    def _lanes(self, v_units: torch.Tensor) -> torch.Tensor:
        """Unit values -> one value per lane row (the joint row's value repeated for its agents)."""
        if self.critic_input == "local":
            return v_units.reshape(-1)
        return v_units.reshape(-1, 1).expand(-1, self.n_agent).reshape(-1)

    def _values(self, obs_rows: torch.Tensor, joint: torch.Tensor | None) -> torch.Tensor:
        """V for every lane row.  local: critic(row).  global: critic(joint row of the env step), repeated per agent."""
        return self._lanes(self._unit_values(obs_rows, joint))

    def _next_values_chained(self, vu_s: torch.Tensor, x_next: torch.Tensor, done: torch.Tensor, T: int) -> torch.Tensor:
        """V(obs_next) per unit [T * U] for T unrotated slots of CHAINED rows (buffer.rows_chained): obs_next of slot t is
        obs of slot t + 1 unless the episode ended at t, so the pass over obs_next (a2c.py:124) repeats the pass over obs
        (a2c.py:123) except for the last slot -- its U rows get their own small pass -- and for rows that end an episode
        early: then (device flag, no host round trip) the full pass runs after all.  Bit-identical to the full pass
        either way; saves half of the preprocess' critic work in the aligned case (collect length == episode length)."""
        U = vu_s.numel() // T
        flag = ops.any_nonzero_u8(done[:T - 1].reshape(-1)) if T > 1 else torch.zeros(1, dtype=torch.int32, device=vu_s.device)
        v_full = self._critic_values(x_next, run_if=flag)
        v_last = self._critic_values(x_next[(T - 1) * U:])
        return ops.value_next_select(vu_s, v_last, v_full, flag, T, U).reshape(-1)

    # ---- rollout side -----------------------------------------------------------------------------------------------
    def act_device(self, obs: torch.Tensor, out: dict | None = None, offset_dev: torch.Tensor | None = None,
                   row_offset: int = 0) -> dict:
        D = self.net.obs_dim
        rows = obs.reshape(-1, D)
        logits = FlatMLP.forward(self.net.actor, rows, save=False)
        joint = obs.reshape(-1, self.n_agent * D) if self.critic_input == "global" else None
        value = self._values(rows, joint)
        greedy = bool(self.deterministic_eval and not self.is_within_training_step)
        res = (out["act"], out["logp"]) if out is not None else None
        act, logp = ops.categorical_sample(logits, self.seed, offset=self._sample_ctr + row_offset, deterministic=greedy,
                                           offset_dev=offset_dev, out=res)
        if offset_dev is None:
            self._sample_ctr += rows.shape[0]
        if out is not None:
            out["value"].copy_(value)
            return out
        return dict(act=act, logp=logp, value=value, logits=logits)

    # ---- update side ------------------------------------------------------------------------------------------------
    def _preprocess_batch(self, buffer: DeviceVectorReplayBuffer, uniform_T: int | None = None,
                          allow_stored: bool = True) -> dict:
        # uniform_T: every sub-buffer holds exactly T unrotated rows (known on the host): no device round trip, so the
        # pass can be captured into a hipGraph
        T, rows, env_start, env_len = (uniform_T, None, None, None) if uniform_T else self._valid_rows(buffer)
        B, N, D = buffer.buffer_num, buffer.n_agent, buffer.obs_dim
        if self.critic_input == "global" and N != self.n_agent:
            raise ValueError(f"buffer holds {N} agents, the centralized critic was built for {self.n_agent}")
        L = B * N
        no_next = buffer.obs_next_store is None  # ignore_obs_next (PPO._next_values_by_index)
        obs = buffer.obs_store[:T].reshape(T * L, D)
        obs_next = None if no_next else buffer.obs_next_store[:T].reshape(T * L, D)
        act = buffer.act_store[:T].reshape(T * L)
        glob = self.critic_input == "global"
        joint = buffer.obs_store[:T].reshape(T * B, N * D) if glob else None
        joint_next = buffer.obs_next_store[:T].reshape(T * B, N * D) if glob and not no_next else None
        reuse = allow_stored and self.reuse_rollout_outputs and rows is None and buffer.logp_store is not None
        # logp_old / v_s produced by the rollout with these very parameters (same kernels, row-wise arithmetic: the same bits
        # as recomputing them, a2c.py:121-127 / ppo.py:157-161) are taken from the buffer
        vu_s = None
        if reuse and buffer.behaviour_outputs_version == self.param_version and not no_next:
            v_s = buffer.vs_store[:T].reshape(T, L)
        else:
            vu_s = self._unit_values(obs, joint)
            v_s = self._lanes(vu_s).view(T, L)
        if reuse and buffer.logp_outputs_version == self.param_version:
            logp_old = buffer.logp_store[:T].reshape(T * L)
        else:
            logp_old, _ = ops.categorical_logp_entropy(FlatMLP.forward(self.net.actor, obs, save=False), act)
        if no_next:
            v_next = self._next_values_by_index(buffer, v_s.contiguous(), T, rows)
        elif vu_s is not None and rows is None and self.shift_next_values and buffer.rows_chained is True:
            v_next = self._lanes(self._next_values_chained(vu_s, joint_next if glob else obs_next, buffer.done_store,
                                                           T)).view(T, L)
        else:
            v_next = self._values(obs_next, joint_next).view(T, L)
        ret, adv = self._gae(v_s, v_next, buffer.rew_store[:T].reshape(T, L), buffer.term_store[:T].reshape(T, L),
                             buffer.trunc_store[:T].reshape(T, L), N, env_start=env_start, env_len=env_len, rows=rows)
        return dict(T=T, rows=rows, obs=obs, act=act, v_s=v_s.reshape(-1).contiguous(), ret=ret.reshape(-1),
                    adv=adv.reshape(-1), logp_old=logp_old, n_env=B, n_agent=N, joint=joint)

    def _grad_step(self, pb: dict, idx: torch.Tensor, adv_stats, step_dev: torch.Tensor | None = None,
                   rows: torch.Tensor | None = None, partial_out: torch.Tensor | None = None) -> torch.Tensor | None:
        """`_grad_step_steps` run to its end, the gradient summed over the data-parallel ranks where it falls due."""
        return drive_steps(self._grad_step_steps(pb, idx, adv_stats, step_dev, rows, partial_out), self._grad_sync)

    def _grad_step_steps(self, pb: dict, idx: torch.Tensor, adv_stats, step_dev: torch.Tensor | None = None,
                         rows: torch.Tensor | None = None, partial_out: torch.Tensor | None = None):
        """One minibatch: forward both nets, loss, backward into joint slabs, clip + Adam.  A generator: with data-parallel
        replicas it YIELDS the flat gradient (scaled by 1 / world) where it has to be summed over the ranks and continues
        with Adam once the caller has reduced it in place (`drive_steps`: one all-reduce; `parallel.learn_lockstep`:
        several policy groups packed into one).  Returns (StopIteration.value) the 4 scalars.
        idx: lane (sample) ids of the minibatch; rows: its joint-row ids when the minibatch is made of whole rows
        (`row_minibatches`: idx == rows * N + agent, row-major).  step_dev: device-resident optimizer step count (graph
        capture); None = the host counter.  partial_out (f64, row-kernel path only): the step's loss partials are left
        there for ONE `ppo_finalize_many` over all steps of the update and None is returned."""
        net = self.net
        if self.fused_actor:
            return (yield from self._grad_step_fused_actor(pb, idx, adv_stats, step_dev, rows, partial_out))
        if rows is not None:
            N = pb["n_agent"]
            cx = ops.gather_rows(pb["joint"], rows)                    # [Mr, N * D]: read once ...
            x = cx.view(-1, net.obs_dim)                               # ... the same bytes are the actor's N rows
            cfg = self._cfg_rows
        else:
            x = ops.gather_rows(pb["obs"], idx)
            cx = x if pb["joint"] is None else ops.gather_rows(pb["joint"], torch.div(idx, pb["n_agent"], rounding_mode="floor"))
            cfg = self._cfg
        logits = FlatMLP.forward(net.actor, x, save=True)
        value = FlatMLP.forward(net.critic, cx, save=True).reshape(-1)
        dlogits, dvalue, scalars = ops.ppo_loss_fwd_bwd(
            logits, value, pb["act"], pb["logp_old"], pb["adv"], pb["ret"], cfg, adv_stats=adv_stats,
            v_s_old=pb["v_s"] if self.value_clip else None, perm=idx)
        M, n_total = idx.numel(), net.flat.numel()
        Mc = value.numel()  # critic rows: M, or M / N with row minibatches
        n_split = ops.mlp_n_split(M)
        slabs = self._ws.get(("slabs", n_split))
        if slabs is None:
            slabs = self._ws[("slabs", n_split)] = torch.empty(n_split, n_total, dtype=torch.float32, device=self.device)
        net.actor.backward(dlogits, n_split, slabs=slabs, slab_stride=n_total)
        net.critic.backward(dvalue.view(Mc, 1), n_split, slabs=slabs[:, net.n_actor:], slab_stride=n_total)
        if step_dev is None:
            self.opt_step += 1
        else:
            ops.call("tsm_u64_add", ops.ptr(step_dev), 1, ops.stream_ptr())
        grads = slabs
        if self._grad_sync is not None:
            flat_g = self._ws.setdefault("flat_grad", torch.empty_like(net.flat.data))
            ops.reduce_slabs(grads, out=flat_g, scale=1.0 / self._grad_sync.world)
            yield flat_g  # summed over the ranks by the driver, in place
            grads = flat_g.view(1, -1)
        ops.adam_step(net.flat.data, grads, self.exp_avg, self.exp_avg_sq, self.opt_step, lr=self.lr, lr_dev=self._lr_dev,
                      betas=self.betas,
                      eps=self.adam_eps, weight_decay=self.weight_decay, max_grad_norm=self.max_grad_norm,
                      work=self._adam_work, step_dev=step_dev)
        return scalars

    def _rows_grids(self, M: int, Mr: int, crit_rows: bool) -> tuple[int, int]:
        """(actor workgroups, loss-partial groups of the value term) of a row-kernel gradient step of M samples / Mr units."""
        return ops.ppo_actor_rows_grid(M), (ops.ppo_critic_rows_grid(Mr) if crit_rows else ops.ppo_loss_partial_elems(M) // 4)

    def _grad_step_fused_actor(self, pb: dict, idx: torch.Tensor, adv_stats, step_dev, rows,
                               partial_out: torch.Tensor | None = None) -> torch.Tensor | None:
        """The same gradient step with the 128-wide actor in ONE launch (forward, policy loss, backward:
        csrc/ppo_rows.hip) and the critic beside it -- in one launch of its own when the minibatch is made of whole joint
        rows or the critic is local, else on the dense GEMMs with the value term alone.  The two halves keep their own
        slab arrays and meet in ONE segmented Adam launch (`tsm_adam_step_segs`: slab sums, the joint gradient norm when
        clipping, Adam); data parallel: one segmented reduction -> all-reduce -> Adam.  Three launches per gradient step
        in the captured update: the optimizer step count is advanced by the actor kernel and the loss statistics of all
        steps are folded by one launch at the end (`partial_out`)."""
        net, dev = self.net, self.device
        M, P_a, P_c = idx.numel(), net.n_actor, net.n_critic
        # the critic in one launch too when the minibatch is made of whole joint rows (or the critic is local)
        crit_rows = self.fused_critic and (rows is not None or pb["joint"] is None)
        Mr = rows.numel() if rows is not None else M
        na, nv = self._rows_grids(M, Mr, crit_rows)
        n_split = nv if crit_rows else ops.mlp_n_split(Mr)
        w = self._ws.get(("rows", M, na, n_split, crit_rows))
        if w is None:
            w = self._ws[("rows", M, na, n_split, crit_rows)] = dict(
                slabs_a=torch.empty(na, P_a, dtype=torch.float32, device=dev),
                slabs_c=None if (crit_rows and self.critic_gen2) else torch.empty(n_split, P_c, dtype=torch.float32, device=dev),
                partial=torch.zeros((na + nv) * 4, dtype=torch.float64, device=dev),
                nb=torch.tensor([na + nv], dtype=torch.int32, device=dev), M=torch.tensor([M], dtype=torch.int64, device=dev),
                flat_g=torch.empty(P_a + P_c, dtype=torch.float32, device=dev),
                # the actor's and the critic's small-gradient slabs summed to one row each inside the dW1 launch (side reductions)
                red_a=torch.empty(1, P_a, dtype=torch.float32, device=dev),
                red_c=torch.empty(1, P_c - net.critic.dims[1] * net.critic_obs_dim, dtype=torch.float32, device=dev))
        partial = w["partial"] if partial_out is None else partial_out
        if partial.numel() < (na + nv) * 4:
            raise ValueError(f"loss partials: {partial.numel()} values, this step needs {(na + nv) * 4}")
        ops.ppo_actor_rows_update(net.actor.flat.data, pb["obs"], pb["act"], pb["logp_old"], pb["adv"], self._cfg, net.n_act,
                                  net.actor.dims[1], adv_stats=adv_stats, perm=idx, M=M, n_blocks=na, slabs=w["slabs_a"],
                                  partial=partial[:na * 4], opt_step_dev=step_dev)
        segs_c = None
        if crit_rows and self.critic_gen2:
            src, N_c = (pb["joint"], pb["n_agent"]) if rows is not None else (pb["obs"], 1)
            nW1 = net.critic.dims[1] * net.critic_obs_dim
            img = self._w1_img if self._grad_sync is None else None  # (kept in step by the segmented Adam launch only)
            split = self.split_dw2
            # side reductions inside the dW1 launch: the actor's slabs always; the critic's small-gradient slabs only while they
            # still carry dW2 (split mode leaves 385 floats per slab: the optimizer reads those directly)
            side = [(w["slabs_a"][:na], w["red_a"][0])] + ([] if split else [("rest", w["red_c"][0])])
            ops.critic_rows_grad_ppo(net.critic.flat.data, src, pb["ret"], self._cfg, N_c, net.critic.dims[1],
                                     v_s_old=pb["v_s"] if self.value_clip else None,
                                     rows=rows if rows is not None else idx, Mr=Mr,
                                     partial=partial[na * 4:(na + nv) * 4], ws=self._ws,
                                     w1_image=img if self._w1_img_ok else None, side_reduce=side, split_dw2=split)
            cw = self._ws[("critic_grad", net.critic_obs_dim, net.critic.dims[1], 1, Mr, False, split)]
            # (the optimizer reads one row per side-reduced segment: the same bits as summing the slabs itself, ops.py)
            segs_c = ops.critic_grad_segs(cw, P_a, net.critic_obs_dim, net.critic.dims[1], 1, w1_image=img,
                                          rest_row=None if split else w["red_c"])
            actor_seg = (w["red_a"], 0, P_a)
        elif crit_rows:
            src, N_c = (pb["joint"], pb["n_agent"]) if rows is not None else (pb["obs"], 1)
            ops.ppo_critic_rows_update(net.critic.flat.data, src, pb["ret"], self._cfg, N_c, net.critic.dims[1],
                                       v_s_old=pb["v_s"] if self.value_clip else None, rows=rows if rows is not None else idx,
                                       Mr=Mr, n_blocks=nv, slabs=w["slabs_c"], partial=partial[na * 4:(na + nv) * 4])
        else:
            if rows is not None:
                cx, vg = ops.gather_rows(pb["joint"], rows), pb["n_agent"]
            elif pb["joint"] is not None:
                cx, vg = ops.gather_rows(pb["joint"], torch.div(idx, pb["n_agent"], rounding_mode="floor")), 1
            else:
                cx, vg = ops.gather_rows(pb["obs"], idx), 1
            value = FlatMLP.forward(net.critic, cx, save=True).reshape(-1)
            dvalue, _ = ops.ppo_value_loss(value, pb["ret"], self._cfg_value[vg], M, v_s_old=pb["v_s"] if self.value_clip else None,
                                           perm=idx, partial=partial[na * 4:(na + nv) * 4])
            net.critic.backward(dvalue.view(-1, 1), n_split, slabs=w["slabs_c"], slab_stride=P_c)
        if step_dev is None:
            self.opt_step += 1
        segs = [actor_seg if crit_rows and self.critic_gen2 else (w["slabs_a"][:na], 0, P_a)] + \
            (segs_c or [(w["slabs_c"][:n_split], P_a, P_c)])
        hyper = dict(lr=self.lr, lr_dev=self._lr_dev, betas=self.betas, eps=self.adam_eps, weight_decay=self.weight_decay,
                     step_dev=step_dev)
        if self._grad_sync is None:
            ops.adam_step_segs(net.flat.data, segs, self.exp_avg, self.exp_avg_sq, self.opt_step,
                               max_grad_norm=self.max_grad_norm, work=self._adam_work, **hyper)
            self._w1_img_ok = segs_c is not None and self._w1_img is not None
        else:
            ops.reduce_slabs_segs(segs, P_a + P_c, out=w["flat_g"], scale=1.0 / self._grad_sync.world)
            yield w["flat_g"]  # summed over the ranks by the driver, in place
            ops.adam_step(net.flat.data, w["flat_g"].view(1, -1), self.exp_avg, self.exp_avg_sq, self.opt_step,
                          max_grad_norm=self.max_grad_norm, work=self._adam_work, **hyper)
        if partial_out is not None:
            return None
        scal = torch.empty(1, 4, dtype=torch.float32, device=dev)
        ops.ppo_finalize_many(w["partial"], (na + nv) * 4, w["nb"], w["M"], self._cfg, scal)
        return scal[0]

    # ---- minibatch plan shared by the graph and the eager path -----------------------------------------------------
    def _plan(self, n_rows: int, N: int, batch_size: int | None):
        """-> (groups, n_g, bounds, unit): `n_g` permuted units per group, `bounds` over units, `unit` = lanes per unit.
        per_agent: units are one agent's lanes; pooled: all lanes; row minibatches: joint rows of N lanes each."""
        if self.dispatch == "per_agent":
            return list(range(N)), n_rows, split_bounds(n_rows, batch_size or -1, merge_last=True), 1
        if self.row_minibatches:
            if batch_size and batch_size > 0 and batch_size % N:
                raise ValueError(f"batch_size={batch_size} must be a multiple of the {N} agents of a joint row "
                                 "(centralized critic, pooled dispatch: minibatches are made of whole rows)")
            return [None], n_rows, split_bounds(n_rows, (batch_size // N) if batch_size and batch_size > 0 else -1,
                                                merge_last=True), N
        return [None], n_rows * N, split_bounds(n_rows * N, batch_size or -1, merge_last=True), 1

    # ---- hipGraph path: one replay per update() (uniform, unrotated buffers; local advantage statistics) ------------
    def _update_graph_generic(self, buffer: DeviceVectorReplayBuffer, batch_size: int | None, repeat: int):
        from ..data.stats import MapTrainingStats

        T = buffer.host_uniform_len()
        if not T:
            return None
        B, N = buffer.buffer_num, buffer.n_agent
        per_agent = self.dispatch == "per_agent"
        groups, n_g, bounds, unit = self._plan(T * B, N, batch_size)
        row_mode = unit > 1
        stored = (self.reuse_rollout_outputs, buffer.behaviour_outputs_version == self.param_version,
                  buffer.logp_outputs_version == self.param_version)
        # every property of the buffer that `_preprocess_batch` branches on at CAPTURE time belongs to the key: a graph
        # captured over chained rows (V(obs_next) from the next slot's V(obs)) must never replay on rows that are not
        chained = buffer.rows_chained is True and self.shift_next_values and buffer.obs_next_store is not None
        key = ("ggraph", buffer.storage_key(), T, batch_size, repeat, self.dispatch, self.max_grad_norm, stored, chained,
               self._grad_sync is not None, self.graph_collectives, ops.kernel_options())
        w = self._ws.get(key)
        if w is None:  # first update of this shape runs eagerly (one-time kernel attributes, allocator warm-up)
            self._ws[key] = {}
            return None
        dev = self.device
        n_steps = len(groups) * repeat * len(bounds)
        if "graph" not in w:
            if self._grad_sync is not None:
                self._grad_sync.require_equal(n_steps, "the number of gradient steps per update")
            w.update(perm=torch.zeros(len(groups), repeat, n_g, dtype=torch.int64, device=dev),
                     step_dev=torch.zeros(1, dtype=torch.int64, device=dev),
                     scal=torch.zeros(n_steps, 4, dtype=torch.float32, device=dev),
                     mb_start=torch.as_tensor([b[0] * unit for b in bounds] + [n_g * unit], dtype=torch.int64, device=dev))
            seg = torch.arange(len(groups) * repeat, dtype=torch.int64, device=dev).view(-1, 1) * (n_g * unit)
            # every tensor a captured kernel reads must outlive the graph: keep it in the workspace
            w["mb_all"] = mb_all = torch.cat([(seg + w["mb_start"][:-1].view(1, -1)).reshape(-1),
                                              torch.tensor([len(groups) * repeat * n_g * unit], dtype=torch.int64, device=dev)])
            w["lane_of_row"] = torch.arange(N, dtype=torch.int64, device=dev).view(1, 1, 1, N)
            # row-kernel steps leave their loss partials behind; ONE launch folds them all at the end of the update
            defer = self.fused_actor
            if defer:
                glob_rows = self.critic_input == "global"
                grids = []
                for s_, e_ in bounds:
                    M_, Mr_ = (e_ - s_) * unit, e_ - s_
                    crit = self.fused_critic and (row_mode or not glob_rows)
                    grids.append(sum(self._rows_grids(M_, Mr_ if row_mode else M_, crit)))
                reps = len(groups) * repeat
                w["partial"] = torch.zeros(n_steps, max(grids) * 4, dtype=torch.float64, device=dev)
                w["nb_dev"] = torch.as_tensor(grids * reps, dtype=torch.int32, device=dev)
                w["M_dev"] = torch.as_tensor([(e_ - s_) * unit for s_, e_ in bounds] * reps, dtype=torch.int64, device=dev)

            def body():
                self._w1_img_ok = False  # (the parameters may have changed since the last step this object took)
                if self.shuffle == "device":
                    ops.random_permutations(n_g, len(groups) * repeat, self.seed ^ 0x5DEECE66D, counter_dev=w["step_dev"],
                                            scale=N if per_agent else 1, group_size=repeat,
                                            offset_mul=1 if per_agent else 0, out=w["perm"])
                # lane ids of every minibatch: the permuted units themselves, or the N lanes of each permuted joint row
                lanes = (w["perm"].unsqueeze(-1) * N + w["lane_of_row"]).reshape(len(groups), repeat, n_g * N) \
                    if row_mode else w["perm"]
                pb = self._preprocess_batch(buffer, uniform_T=T)
                stats = None
                if self.advantage_normalization:
                    stats = ops.ppo_adv_stats(pb["adv"], mb_all, perm=lanes.reshape(-1),
                                              max_rows=max(e - s for s, e in bounds) * unit)
                    stats = self._global_adv_stats(stats, mb_all).view(len(groups), repeat, len(bounds), 2)
                k = 0
                for gi in range(len(groups)):
                    for r in range(repeat):
                        for j, (s, e) in enumerate(bounds):
                            sc = self._grad_step(pb, lanes[gi, r, s * unit:e * unit], None if stats is None else stats[gi, r, j],
                                                 step_dev=w["step_dev"], rows=w["perm"][gi, r, s:e] if row_mode else None,
                                                 partial_out=w["partial"][k] if defer else None)
                            if not defer:
                                w["scal"][k].copy_(sc)
                            k += 1
                if defer:  # the loss statistics of every gradient step: one launch
                    ops.ppo_finalize_many(w["partial"], w["partial"].shape[1], w["nb_dev"], w["M_dev"], self._cfg, w["scal"])

            graph = torch.cuda.CUDAGraph()
            self._capture_graph(graph, body)  # (data parallel over RCCL: the all-reduces of every step are captured too)
            w["graph"] = graph
            if self.shuffle == "numpy":
                base = ref_order_rows(T, B, dev)
                w["ref_ids"] = [base if row_mode else base * N + a if a is not None else
                                (base[:, None] * N + torch.arange(N, device=dev)[None, :]).reshape(-1) for a in groups]
        if self.shuffle == "numpy":
            for gi, a in enumerate(groups):
                for r in range(repeat):
                    pl = torch.as_tensor(np.random.permutation(n_g)).to(dev)
                    w["perm"][gi, r].copy_(w["ref_ids"][gi][pl])  # reference batch position -> lane id (ppo.ref_order_rows)
        if w.get("step_host") != self.opt_step:
            w["step_dev"].fill_(self.opt_step)
        w["graph"].replay()
        self.opt_step += n_steps
        w["step_host"] = self.opt_step
        self.param_version += 1
        mk = lambda x: A2CTrainingStats(  # noqa: E731
            loss=SequenceSummaryStats.from_sequence(x[:, 0]), actor_loss=SequenceSummaryStats.from_sequence(x[:, 1]),
            vf_loss=SequenceSummaryStats.from_sequence(x[:, 2]), ent_loss=SequenceSummaryStats.from_sequence(x[:, 3]),
            gradient_steps=len(x))

        def finish(s_h):
            if per_agent:
                per = len(s_h) // N
                return MapTrainingStats({f"agent_{a}": mk(s_h[a * per:(a + 1) * per]) for a in range(N)})
            return mk(s_h)

        sync = self._grad_sync
        if sync is not None:
            sync.post_check()  # (peer-memory all-reduce: a lost peer is reported where the statistics are read)
        if not self.async_stats:
            s_h = w["scal"].cpu().numpy()
            if sync is not None:
                sync.raise_if_failed()
            return finish(s_h)
        # async_stats=True (as PPO.update): the statistics travel to a pinned host slot behind the replay and are parsed when
        # the returned object is first read -- the host goes on to queue the next collect while the update runs; a ring of
        # four slots bounds the run-ahead
        ring = w.setdefault("ring", [])
        if len(ring) < 4:
            ring.append(dict(h=torch.empty(w["scal"].shape, dtype=torch.float32, pin_memory=True), event=torch.cuda.Event(),
                             pending=None))
            slot = ring[-1]
        else:
            slot = ring[w.get("ring_pos", 0) % 4]
            w["ring_pos"] = w.get("ring_pos", 0) + 1
            if slot["pending"] is not None:
                slot["pending"].expire("training stats were not read within 4 update() calls (async_stats=True)")
            slot["event"].synchronize()
        slot["h"].copy_(w["scal"], non_blocking=True)
        slot["event"].record()

        def build():
            slot["event"].synchronize()
            if sync is not None:
                sync.raise_if_failed()
            slot["pending"] = None
            return finish(slot["h"].numpy().copy())

        from ..data.stats import LazyStats

        out = LazyStats(build)
        slot["pending"] = out
        return out

    def _update(self, buffer: DeviceVectorReplayBuffer, batch_size: int | None, repeat: int, t0: float):
        # (data parallel: only with capturable collectives -- RCCL; otherwise eager launches with inline collectives)
        if self.graph and (self._grad_sync is None or self.graph_collectives) and not self.recompute_adv:
            out = self._update_graph_generic(buffer, batch_size, repeat)
            if out is not None:
                return out
        return super()._update(buffer, batch_size, repeat, t0)

    def _update_with_batch(self, pb: dict, batch_size: int | None, repeat: int, agent: int | None = None,
                           buffer: DeviceVectorReplayBuffer | None = None, perm_base: int | None = None) -> A2CTrainingStats:
        return drive_steps(self._update_steps(pb, batch_size, repeat, agent=agent, buffer=buffer, perm_base=perm_base),
                           self._grad_sync)

    def _update_steps(self, pb: dict, batch_size: int | None, repeat: int, agent: int | None = None,
                      buffer: DeviceVectorReplayBuffer | None = None, perm_base: int | None = None):
        """The minibatch loop as a generator of gradient synchronisation points (see PPO._update_steps)."""
        dev = self.device
        N = pb["n_agent"]
        row_mode = agent is None and self.row_minibatches and pb.get("joint") is not None and N > 1
        if row_mode:  # units = joint rows in the reference's sample(0) order (or the ragged row list)
            if pb["rows"] is not None:
                ids = pb["rows"]
            elif self.shuffle == "numpy":
                ids = ref_order_rows(pb["T"], pb["n_env"], dev)
            else:
                ids = torch.arange(pb["T"] * pb["n_env"], dtype=torch.int64, device=dev)
            if batch_size and batch_size > 0 and batch_size % N:
                raise ValueError(f"batch_size={batch_size} must be a multiple of the {N} agents of a joint row")
            unit, size = N, (batch_size // N) if batch_size and batch_size > 0 else -1
        else:
            ids = self._sample_ids(pb, agent)
            if ids is None:
                ids = torch.arange(pb["obs"].shape[0], dtype=torch.int64, device=dev)
            unit, size = 1, batch_size or -1
        n = ids.numel()
        bounds = split_bounds(n, size, merge_last=True)
        if self._grad_sync is not None:  # (a collective on every eager update: parallel.GradSync.require_equal)
            self._grad_sync.require_equal(repeat * len(bounds), "the number of gradient steps per update")
        mb_start = torch.as_tensor([b[0] * unit for b in bounds] + [n * unit], dtype=torch.int64, device=dev)
        lane_of_row = torch.arange(N, dtype=torch.int64, device=dev).view(1, N)
        scal = []
        self._w1_img_ok = False  # (the parameters may have changed since the last step this object took)
        for step in range(repeat):
            if self.recompute_adv and step > 0:
                pb = dict(self._preprocess_batch(buffer, allow_stored=False), logp_old=pb["logp_old"])
            if self.shuffle == "numpy":
                perm_local = torch.as_tensor(np.random.permutation(n)).to(dev)
            else:
                perm_local = self._device_perm(n, perm_base, agent, repeat, step)
            perm = ids[perm_local]
            lanes = (perm.view(-1, 1) * N + lane_of_row).reshape(-1) if row_mode else perm
            stats = (ops.ppo_adv_stats(pb["adv"], mb_start, perm=lanes, max_rows=max(e - s for s, e in bounds) * unit)
                     if self.advantage_normalization else None)
            self._global_adv_stats(stats, mb_start)
            for j, (s, e) in enumerate(bounds):
                sc = yield from self._grad_step_steps(pb, lanes[s * unit:e * unit].contiguous(),
                                                      None if stats is None else stats[j],
                                                      rows=perm[s:e].contiguous() if row_mode else None)
                scal.append(sc)
        self.param_version += 1
        if self._grad_sync is not None:
            self._grad_sync.post_check()
        s_h = torch.stack(scal).cpu().numpy()
        if self._grad_sync is not None:
            self._grad_sync.raise_if_failed()
        return A2CTrainingStats(
            loss=SequenceSummaryStats.from_sequence(s_h[:, 0]), actor_loss=SequenceSummaryStats.from_sequence(s_h[:, 1]),
            vf_loss=SequenceSummaryStats.from_sequence(s_h[:, 2]), ent_loss=SequenceSummaryStats.from_sequence(s_h[:, 3]),
            gradient_steps=len(scal))

    def learn(self, batch: Batch, batch_size: int | None = None, repeat: int = 1, **kwargs) -> dict[str, float]:
        """One PPO pass on an explicit agent batch (training_coordinator.py:336); a centralized critic takes
        `batch.global_obs` / `batch.global_obs_next`.  With `graph=True`: the call's launch sequence on static buffers --
        ONE hipGraph replay per call from the second call of a shape on; for a data-parallel replica with its collectives
        captured (RCCL) or between segmented graphs (`parallel.learn_lockstep_graph`), as `PPO.learn`."""
        if self.learn_graph_ok(repeat):
            if self._grad_sync is None:
                return self._learn_graph(batch, batch_size, repeat)
            from ..parallel import learn_lockstep_graph, lockstep_graphs_enabled

            if lockstep_graphs_enabled():
                return learn_lockstep_graph([(self, batch, batch_size, repeat)], self._grad_sync)[0]
        return drive_steps(self.learn_steps(batch, batch_size, repeat, **kwargs), self._grad_sync)

    def learn_graph_ok(self, repeat: int = 1) -> bool:
        """Can `learn` run from static buffers inside captured graphs?  (`graph` is this class's capture switch; recompute_advantage
        re-runs the critic between repeats from the host.)"""
        return bool(self.graph and not (self.recompute_adv and repeat > 1))

    def _learn_static(self, n: int, batch_size: int | None, repeat: int, has_trunc: bool) -> dict:
        """`PPO._learn_static` for the wide nets: static HBM buffers + `w["body"]`, a generator that issues `learn_steps`' launches
        (critic passes, log-probabilities, GAE, permutations, advantage statistics, every gradient step) on them -- the same
        launches in the same order, hence the same bits -- and yields every tensor that has to be summed over the ranks where
        `learn_steps` reduces it.  Optimizer step count and permutation counter live in HBM.  `w["warm"]`: the first call of a
        shape runs the body eagerly (one-time kernel attributes, workspaces), later calls capture / replay."""
        dev, net = self.device, self.net
        D, Kc, glob = net.obs_dim, net.critic_obs_dim, self.critic_input == "global"
        dp = self._grad_sync is not None
        key = ("glearn_graph", n, batch_size, repeat, has_trunc, self.shuffle, dp, ops.kernel_options())
        w = self._ws.get(key)
        if w is not None:
            return w
        # (callers whose batch length changes from call to call -- n_episode collection -- would otherwise grow one set of static
        # buffers and one graph per length without bound: keep the eight most recent shapes)
        old = [k for k in self._ws if isinstance(k, tuple) and k and k[0] == "glearn_graph"]
        for k in old[:-7]:
            gone = self._ws.pop(k)
            if self._grad_sync is not None:
                # the captured lock-step pins every policy's static buffers (parallel.learn_lockstep_graph keys its graphs by
                # id(w)): drop the graphs that replay into the evicted set, or the eviction frees nothing
                cache = self._grad_sync.__dict__.get("_lockstep_graphs", {})
                for ck in [ck for ck in cache if any(wid == id(gone) for _, wid in ck[0])]:
                    del cache[ck]
        bounds = split_bounds(n, batch_size or -1, merge_last=True)
        n_steps = repeat * len(bounds)
        z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device=dev)  # noqa: E731
        w = dict(n=n, n_steps=n_steps, repeat=repeat, has_trunc=has_trunc, warm=False,
                 obs=z(n, D), obs_next=z(n, D), act=z(n, dt=torch.int32), rew=z(n, 1), term=z(n, 1, dt=torch.uint8),
                 trunc=z(n, 1, dt=torch.uint8), scal=z(n_steps, 4), step_dev=z(1, dt=torch.int64), perm=z(repeat, n, dt=torch.int64),
                 mb_start=torch.as_tensor([b[0] for b in bounds] + [n], dtype=torch.int64, device=dev))
        if glob:
            w.update(joint=z(n, Kc), joint_next=z(n, Kc))
        defer = self.fused_actor  # row-kernel steps leave their loss partials behind: ONE launch folds them all (as update())
        if defer:
            grids = [sum(self._rows_grids(e - s, e - s, self.fused_critic and not glob)) for s, e in bounds]
            w.update(partial=z(n_steps, max(grids) * 4, dt=torch.float64),
                     nb_dev=torch.as_tensor(grids * repeat, dtype=torch.int32, device=dev),
                     M_dev=torch.as_tensor([e - s for s, e in bounds] * repeat, dtype=torch.int64, device=dev))

        def body():
            self._w1_img_ok = False  # (the parameters may have changed since the last step this object took)
            obs, act = w["obs"], w["act"]
            if glob:
                joint = w["joint"]
                v_s, v_next = self._critic_values(joint), self._critic_values(w["joint_next"])
            else:
                joint = None
                v_s, v_next = self._critic_values(obs), self._critic_values(w["obs_next"])
            logp_old, _ = ops.categorical_logp_entropy(FlatMLP.forward(net.actor, obs, save=False), act)
            ret, adv = ops.gae_lanes(v_s.view(n, 1), v_next.view(n, 1), w["rew"], w["term"], w["trunc"], self.gamma, self.gae_lambda)
            pb = dict(T=n, rows=None, obs=obs, act=act, v_s=v_s.contiguous(), ret=ret.reshape(-1), adv=adv.reshape(-1),
                      logp_old=logp_old, n_env=1, n_agent=1, joint=joint)
            k = 0
            for r in range(repeat):
                if self.shuffle != "numpy":
                    ops.random_permutations(n, 1, self.seed ^ 0x5DEECE66D, counter_dev=self._perm_ctr, out=w["perm"][r:r + 1])
                    ops.call("tsm_u64_add", ops.ptr(self._perm_ctr), 1, ops.stream_ptr())
                perm = w["perm"][r]
                stats = (ops.ppo_adv_stats(pb["adv"], w["mb_start"], perm=perm, max_rows=max(e - s for s, e in bounds))
                         if self.advantage_normalization else None)
                yield from self._global_adv_stats_steps(stats, w["mb_start"])
                for j, (s_, e_) in enumerate(bounds):
                    sc = yield from self._grad_step_steps(pb, perm[s_:e_], None if stats is None else stats[j],
                                                          step_dev=w["step_dev"], partial_out=w["partial"][k] if defer else None)
                    if not defer:
                        w["scal"][k].copy_(sc)
                    k += 1
            if defer:
                ops.ppo_finalize_many(w["partial"], w["partial"].shape[1], w["nb_dev"], w["M_dev"], self._cfg, w["scal"])

        w["body"] = body
        self._ws[key] = w
        return w

    def _learn_load(self, w: dict, batch: Batch) -> None:
        super()._learn_load(w, batch)
        if "joint" in w:
            if "global_obs" not in batch:
                raise ValueError("GenericPPO(critic_input='global').learn needs batch.global_obs / global_obs_next")
            t = lambda x: x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))  # noqa: E731
            w["joint"].copy_(t(batch.global_obs).reshape(w["joint"].shape), non_blocking=True)
            w["joint_next"].copy_(t(batch.global_obs_next).reshape(w["joint"].shape), non_blocking=True)

    def _learn_graph(self, batch: Batch, batch_size: int | None, repeat: int) -> dict[str, float]:
        """`learn` of a single replica-less policy: eager launches on the static buffers at the first call of a shape, ONE
        hipGraph replay from the second on."""
        w = self._learn_static(len(batch.rew), batch_size, repeat, "truncated" in batch)
        self._learn_load(w, batch)
        if not w["warm"]:
            for _ in w["body"]():
                raise RuntimeError("a single-GPU learn() has no collectives")
            w["warm"] = True
            return self._learn_finish(w)
        if "graph" not in w:
            graph = torch.cuda.CUDAGraph()
            with ops.graph_capture(graph):
                for _ in w["body"]():
                    raise RuntimeError("a single-GPU learn() has no collectives")
            w["graph"] = graph
        w["graph"].replay()
        return self._learn_finish(w)

    def learn_steps(self, batch: Batch, batch_size: int | None = None, repeat: int = 1, **kwargs):
        """`learn` as a generator of gradient synchronisation points: wide nets take part in the lock-step of grouped /
        league policies under data parallelism (`parallel.learn_lockstep`)."""
        dev = self.device
        t = lambda x, dt: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))).to(dev, dt).contiguous()  # noqa: E731
        obs, obs_next = t(batch.obs, torch.float32), t(batch.obs_next, torch.float32)
        n = obs.shape[0]
        act = t(batch.act, torch.int32).reshape(n)
        if self.critic_input == "global":
            if "global_obs" not in batch:
                raise ValueError("GenericPPO(critic_input='global').learn needs batch.global_obs / global_obs_next")
            joint, joint_next = t(batch.global_obs, torch.float32), t(batch.global_obs_next, torch.float32)
            v_s, v_next = self._critic_values(joint), self._critic_values(joint_next)
        else:
            joint = None
            v_s, v_next = self._critic_values(obs), self._critic_values(obs_next)
        logp_old, _ = ops.categorical_logp_entropy(FlatMLP.forward(self.net.actor, obs, save=False), act)
        term = t(batch.terminated, torch.uint8).reshape(n, 1)
        trunc = t(batch.truncated, torch.uint8).reshape(n, 1) if "truncated" in batch else torch.zeros_like(term)
        ret, adv = ops.gae_lanes(v_s.view(n, 1), v_next.view(n, 1), t(batch.rew, torch.float32).view(n, 1), term, trunc,
                                 self.gamma, self.gae_lambda)
        pb = dict(T=n, rows=None, obs=obs, act=act, v_s=v_s.contiguous(), ret=ret.reshape(-1), adv=adv.reshape(-1),
                  logp_old=logp_old, n_env=1, n_agent=1, joint=joint)
        st = yield from self._update_steps(pb, batch_size, repeat)
        return {"loss": st.loss.mean, "actor_loss": st.actor_loss.mean, "vf_loss": st.vf_loss.mean,
                "ent_loss": st.ent_loss.mean}

    # ---- snapshots (checkpoints: PPO.state_dict / load_state_dict on the reference key names) ------------------
    def __deepcopy__(self, memo):
        src = self.net
        net = MLPActorCritic(src.obs_dim, src.n_act, tuple(src.actor.dims[1:-1]), act=src.actor.act,
                             critic_obs_dim=src.critic_obs_dim, device=self.device)
        net._ref_keys = getattr(src, "_ref_keys", None)
        ctor = copy.copy(self._ctor)
        new = GenericPPO(net=net, **ctor)
        new.load_state_dict(self.state_dict())
        new.train(self.training)
        return new
