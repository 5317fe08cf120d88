"""A2C and Reinforce on the device-resident rollout: the plain policy-gradient members of the reference's on-policy
family, running on the SAME kernels as PPO.

    A2C        /root/reference/tianshou/algorithm/modelfree/a2c.py:152-285
               loss = -mean(log_prob * adv) + vf_coef * mse(returns, value) - ent_coef * mean(entropy);
               advantages / returns from the same `_add_returns_and_advantages` as PPO (a2c.py:113-151), no
               advantage normalisation, no clipping, no recompute.
    Reinforce  /root/reference/tianshou/algorithm/modelfree/reinforce.py:240-379
               loss = -mean(log_prob * returns); returns = discounted Monte-Carlo returns
               (`compute_episodic_return` with gae_lambda = 1 and v_s_ = ret_rms.mean, reinforce.py:294-311),
               optionally standardised with a RunningMeanStd that is updated AFTER use.

Both reuse PPO's machinery (buffer views, permutations, per-agent dispatch, fused forward+loss+backward kernel with
`tsm_ppo_cfg.loss_kind = 1`, slab reduction + Adam, hipGraph replay for A2C) -- only the loss configuration and, for
Reinforce, the return computation differ.  The fused network layout always carries a critic; Reinforce leaves it
untouched (zero gradient => Adam does not move it).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from ..data.buffer import DeviceVectorReplayBuffer
from ..data.stats import SequenceSummaryStats, TrainingStats
from ..utils.net import DiscreteActorCritic
from .ppo import PPO


class A2C(PPO):
    """Synchronous advantage actor-critic (a2c.py:152-285) with the reference's constructor arguments."""

    def __init__(self, *, policy=None, critic=None, optim=None, vf_coef: float = 0.5, ent_coef: float = 0.01,
                 max_grad_norm: float | None = None, gae_lambda: float = 0.95, max_batchsize: int = 256,
                 gamma: float = 0.99, return_scaling: bool = False, net: DiscreteActorCritic | None = None, **kw) -> None:
        for banned in ("eps_clip", "dual_clip", "value_clip", "advantage_normalization", "recompute_advantage"):
            if banned in kw:
                raise TypeError(f"A2C has no `{banned}` (that is a PPO option)")
        super().__init__(policy=policy, critic=critic, optim=optim, net=net, vf_coef=vf_coef, ent_coef=ent_coef,
                         max_grad_norm=max_grad_norm, gae_lambda=gae_lambda, max_batchsize=max_batchsize, gamma=gamma, return_scaling=return_scaling,
                         advantage_normalization=False, recompute_advantage=False, value_clip=False, dual_clip=None, **kw)
        self._cfg = ops.make_ppo_cfg(adv_norm=False, vf_coef=vf_coef, ent_coef=ent_coef, loss_kind=1)
        self._a2c_ctor = dict(vf_coef=vf_coef, ent_coef=ent_coef, max_grad_norm=max_grad_norm, gae_lambda=gae_lambda,
                              max_batchsize=max_batchsize, gamma=gamma, return_scaling=return_scaling, **kw)

    def __deepcopy__(self, memo):
        net = DiscreteActorCritic(self.net.obs_dim, self.net.n_act, self.net.hidden, device=self.device)
        net.flat.data.copy_(self.net.flat.data)
        net.sync_image()
        net._ref_keys = getattr(self.net, "_ref_keys", None)
        new = type(self)(net=net, **self._a2c_ctor)
        new.load_state_dict(self.state_dict())
        new.train(self.training)
        return new


class LossSequenceTrainingStats(TrainingStats):
    """reinforce.py:59-60: the only statistic of Reinforce is the loss sequence."""

    _non_loss_fields = TrainingStats._non_loss_fields + ("gradient_steps",)

    def __init__(self, loss: SequenceSummaryStats, gradient_steps: int = 0) -> None:
        super().__init__()
        self.loss = loss
        self.gradient_steps = gradient_steps


class Reinforce(PPO):
    """Vanilla policy gradient (reinforce.py:313-379).  `return_standardization` as in
    DiscountedReturnComputation (reinforce.py:240-311)."""

    def __init__(self, *, policy=None, gamma: float = 0.99, return_standardization: bool = False, optim=None,
                 net: DiscreteActorCritic | None = None, **kw) -> None:
        for banned in ("eps_clip", "dual_clip", "value_clip", "advantage_normalization", "recompute_advantage",
                       "vf_coef", "ent_coef", "gae_lambda", "return_scaling", "max_grad_norm"):
            if banned in kw:
                raise TypeError(f"Reinforce has no `{banned}`")
        kw.pop("use_graph", None)
        # the return standardisation keeps running statistics on the host: the update stays on eager launches
        super().__init__(policy=policy, critic=None, optim=optim, net=net, gamma=gamma, gae_lambda=1.0, vf_coef=0.0,
                         ent_coef=0.0, advantage_normalization=False,
                         recompute_advantage=False, value_clip=False, dual_clip=None, return_scaling=False,
                         use_graph=False, **kw)
        self.return_standardization = bool(return_standardization)
        self._cfg = ops.make_ppo_cfg(adv_norm=False, vf_coef=0.0, ent_coef=0.0, loss_kind=1)
        self._r_ctor = dict(gamma=gamma, return_standardization=return_standardization, **kw)

    def _preprocess_batch(self, buffer: DeviceVectorReplayBuffer) -> dict:
        """reinforce.py:273-311: returns = discounted Monte-Carlo return, bootstrapped with ret_rms.mean where an
        episode is cut without terminating.  The reference feeds `_gae` v_s_ = mean * value_mask and v_s =
        roll(v_s_, 1); with lambda = 1 the v_s terms telescope inside an episode, so the same returns come from the
        lane scan with v_s = v_s_ = mean (the kernel applies the value mask and the end flags itself)."""
        T, rows, env_start, env_len = self._valid_rows(buffer)
        B, N, D = buffer.buffer_num, buffer.n_agent, buffer.obs_dim
        L = B * N
        dev = self.device
        c = torch.full((T, L), float(self.ret_rms.mean), dtype=torch.float32, device=dev)
        ret, _ = ops.gae_lanes(c, c, buffer.rew_store[:T].reshape(T, L), buffer.term_store[:T].reshape(T, L),
                               buffer.trunc_store[:T].reshape(T, L), self.gamma, 1.0, lanes_per_env=N,
                               env_start=env_start, env_len=env_len)
        used = ret
        if self.return_standardization:
            used = (ret - float(self.ret_rms.mean)) / float(np.sqrt(self.ret_rms.var + self._eps))
            sel = ret.view(T * B, N)[rows] if rows is not None else ret
            self.ret_rms.update(sel)  # reinforce.py:305-309: statistics are updated after they were used
        zeros = torch.zeros(T * L, dtype=torch.float32, device=dev)
        flat = used.reshape(-1).contiguous()
        return dict(T=T, rows=rows, obs=buffer.obs_store[:T].reshape(T * L, D), act=buffer.act_store[:T].reshape(T * L),
                    v_s=zeros, ret=flat, adv=flat, logp_old=zeros, n_env=B, n_agent=N)

    def _update_with_batch(self, pb, batch_size, repeat, agent=None, buffer=None, perm_base=None):
        st = super()._update_with_batch(pb, batch_size, repeat, agent=agent, buffer=buffer, perm_base=perm_base)
        return LossSequenceTrainingStats(loss=st.loss, gradient_steps=st.gradient_steps)

    # `learn` below has its own preprocessing (no critic, Monte-Carlo returns): PPO's static-buffer / generator forms of
    # learn() do not describe it, so Reinforce policies of a data-parallel manager learn one after the other
    learn_steps = None

    def learn_graph_ok(self, repeat: int = 1) -> bool:
        return False

    def learn(self, batch, batch_size: int | None = None, repeat: int = 1, **kwargs) -> dict[str, float]:
        """One Reinforce pass on an explicit agent batch (one time-ordered lane; the last row ends the lane)."""
        dev = self.device
        t = lambda x, dt: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))).to(dev, dt).contiguous()  # noqa: E731
        obs = t(batch.obs, torch.float32)
        n = obs.shape[0]
        term = t(batch.terminated, torch.uint8).reshape(n, 1)
        trunc = t(batch.truncated, torch.uint8).reshape(n, 1) if "truncated" in batch else torch.zeros_like(term)
        c = torch.full((n, 1), float(self.ret_rms.mean), dtype=torch.float32, device=dev)
        ret, _ = ops.gae_lanes(c, c, t(batch.rew, torch.float32).view(n, 1), term, trunc, self.gamma, 1.0)
        used = ret
        if self.return_standardization:
            used = (ret - float(self.ret_rms.mean)) / float(np.sqrt(self.ret_rms.var + self._eps))
            self.ret_rms.update(ret)
        flat = used.reshape(-1).contiguous()
        zeros = torch.zeros(n, dtype=torch.float32, device=dev)
        self.net.sync_image()
        pb = dict(T=n, rows=None, obs=obs, act=t(batch.act, torch.int32).reshape(n), v_s=zeros, ret=flat, adv=flat,
                  logp_old=zeros, n_env=1, n_agent=1)
        st = self._update_with_batch(pb, batch_size, repeat)
        return {"loss": st.loss.mean}

    def __deepcopy__(self, memo):
        net = DiscreteActorCritic(self.net.obs_dim, self.net.n_act, self.net.hidden, device=self.device)
        net.flat.data.copy_(self.net.flat.data)
        net.sync_image()
        net._ref_keys = getattr(self.net, "_ref_keys", None)
        new = type(self)(net=net, **self._r_ctor)
        new.load_state_dict(self.state_dict())
        new.train(self.training)
        return new
