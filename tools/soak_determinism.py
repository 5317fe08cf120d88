#!/usr/bin/env python3
"""Soak + determinism check of the headline job: run N training steps (collect + update) and print a hash of the final
parameters, optimizer state and buffer contents.  Two runs with the same seeds must print the same line: every kernel
on the path is deterministic (fixed-order reductions, counter-based RNG), so a data race would show up as a mismatch.

    python tools/soak_determinism.py [n_steps] [--stable | --c3 [--noclip] | --ctde | --tag]

--stable: gamma = 0.95, max_grad_norm = 0.5 -- the configuration under which the job keeps learning (with the reference's
defaults the critic diverges after ~600 updates), so that the hash covers a policy that learns.
--c3 / --tag: the same check for the BASELINE configs[2] job (rows kernels, actor rollout) and the configs[4] shard
(two-team rollout, league trainer), both at gamma = 0.95, max_grad_norm = 0.5.
--c3 --noclip: the C3 job without a gradient-norm clip and with async statistics -- the path on which one segmented Adam launch
reads the actor / critic / dW1 slabs directly (round 3).  --ctde: the reference's CTDEPolicy job (`bench.py --workload c3`):
eight learn() calls per step on the buffer's stores, one hipGraph replay each (round 3).
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class A:
    n_env, n_agent, horizon, minibatch, repeat, dispatch = 1024, 3, 25, 4096, 1, "per_agent"


def _hash(tensors) -> str:
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def soak_c3(n: int) -> None:
    """BASELINE configs[2]: 4096 envs x 8 agents, actor-only persistent rollout + row-minibatch PPO with a centralized
    critic (rows kernels, chained next values, one hipGraph per update)."""
    from tianshou_marl_amd.algorithm import GenericPPO, policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.utils.net import MLPActorCritic

    dev, n_env, N, T = "cuda", 4096, 8, 25
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=dev, seed=1626)
    net = MLPActorCritic(env.obs_dim, 5, (128, 128), critic_obs_dim=N * env.obs_dim, device=dev, seed=1626)
    noclip = "--noclip" in sys.argv
    algo = GenericPPO(net=net, critic_input="global", n_agent=N, lr=3e-4, gamma=0.95, max_grad_norm=None if noclip else 0.5,
                      shuffle="device", seed=1626, dispatch="pooled", async_stats=noclip)
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=dev)
    col = Collector(algo, env, buf)
    col.reset()
    for i in range(n):
        with policy_within_training_step(algo):
            cs = col.collect(n_step=n_env * T)
            algo.update(buf, 65536, 1)
        col.reset_buffer(keep_statistics=True)
        if i % 100 == 0:
            print(f"step {i}: mean episode return {float(cs.returns.mean()):.3f}", flush=True)
    torch.cuda.synchronize()
    print(f"c3{' noclip async' if noclip else ''} steps {n} opt_step {algo.opt_step} sha256 "
          f"{_hash((net.flat.data, algo.exp_avg, algo.exp_avg_sq, buf.obs_store, buf.act_store, buf.rew_store, buf.logp_store, env.agent_pos))}")


def soak_ctde(n: int) -> None:
    """The reference's CTDE job at configs[2]'s shape: shared 128-wide actor, centralized critic with 8 outputs, every agent's
    learn() per step reading the stores in place (critic_rows / critic_train / critic_dw1 / actor rows kernels, two Adam
    launches), one hipGraph replay per call."""
    from tianshou_marl_amd.algorithm.multiagent import (CentralizedCritic, CTDEPolicy, DecentralizedActor,
                                                        FlexibleMultiAgentPolicyManager, agent_batches_from_buffer)
    from tianshou_marl_amd.algorithm.multiagent.training_coordinator import _attach_global
    from tianshou_marl_amd.algorithm.ppo import policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv

    dev, n_env, N, T = "cuda", 4096, 8, 25
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=dev, seed=1626)
    D = env.obs_dim
    pol = CTDEPolicy(actor=DecentralizedActor(D, 5, 128, device=dev, seed=1626), critic=CentralizedCritic(N * D, N, 128, device=dev, seed=1627),
                     seed=1626, async_stats=True, discount_factor=0.95)
    mgr = FlexibleMultiAgentPolicyManager(pol, env, mode="shared")
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, D, device=dev)
    col = Collector(mgr, env, buf)
    col.reset()
    last = None
    for i in range(n):
        with policy_within_training_step(mgr):
            cs = col.collect(n_step=n_env * T)
            batches = agent_batches_from_buffer(buf, env.agents, copies=False)
            for a in env.agents:
                last = pol.learn(_attach_global(batches, batches[a]))
        col.reset_buffer(keep_statistics=True)
        if i % 50 == 0:
            print(f"step {i}: mean episode return {float(cs.returns.mean()):.3f} critic_loss {float(last['critic_loss']):.4f}", flush=True)
    torch.cuda.synchronize()
    print(f"ctde steps {n} opt_steps {pol.optim_actor.step_count} / {pol.optim_critic.step_count} sha256 "
          f"{_hash((pol.actor.flat.data, pol.critic.flat.data, pol.optim_actor.exp_avg_sq, pol.optim_critic.exp_avg_sq, buf.obs_store, buf.act_store, buf.rew_store, env.agent_pos))}")


def soak_tag(n: int) -> None:
    """BASELINE configs[4] shard: 512 simple_tag worlds, one policy per team (two-team persistent rollout), league trainer
    (learn() as one graph replay per learner, time-parallel GAE)."""
    import numpy as np

    from tianshou_marl_amd.algorithm.multiagent import FlexibleMultiAgentPolicyManager, LeaguePlayTrainer, agent_batches_from_buffer
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe_tag import DeviceSimpleTagVectorEnv
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

    dev, n_env, T = "cuda", 512, 25
    env = DeviceSimpleTagVectorEnv(n_env, device=dev, seed=1626, max_cycles=T)
    mk = lambda s_: PPO(net=DiscreteActorCritic(env.obs_dim, 5, 64, device=dev, seed=s_), seed=s_, lr=3e-4, gamma=0.95,  # noqa: E731
                        max_grad_norm=0.5, shuffle="device")
    teams = {"adversaries": mk(1626), "good": mk(1627)}
    mgr = FlexibleMultiAgentPolicyManager(teams, env, mode="grouped", agent_groups=env.agent_groups)
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, env.n_agent, env.obs_dim, device=dev)
    col = Collector(mgr, env, buf)
    col.reset()
    trainer = LeaguePlayTrainer(mgr, matchmaking="random")
    np.random.seed(1626)
    for i in range(n):
        with policy_within_training_step(mgr):
            cs = col.collect(n_step=n_env * T)
            batch = agent_batches_from_buffer(buf, env.agents, only=["agent_0", "adversary_0"], global_state=False)
            batch["good"], batch["adversaries"] = batch["agent_0"], batch["adversary_0"]
            trainer.train_step(batch)
        col.reset_buffer(keep_statistics=True)
        if i % 200 == 0:
            r = cs.returns.mean(0)
            print(f"step {i}: mean episode return adversary_0 {float(r[0]):.2f} agent_0 {float(r[-1]):.2f}", flush=True)
    torch.cuda.synchronize()
    ts = [p.net.flat.data for p in teams.values()] + [p.exp_avg_sq for p in teams.values()] + \
         [buf.obs_store, buf.act_store, buf.rew_store, buf.logp_store, buf.vs_store, env.agent_pos, env.landmark_pos]
    print(f"tag steps {n} opt_steps {[p.opt_step for p in teams.values()]} sha256 {_hash(ts)}")


def main():
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    n = int(args[0]) if args else 5000
    if "--c3" in sys.argv:
        return soak_c3(n)
    if "--tag" in sys.argv:
        return soak_tag(n)
    if "--ctde" in sys.argv:
        return soak_ctde(n)
    a = A()
    if "--stable" in sys.argv:
        a.ppo_kwargs = dict(gamma=0.95, max_grad_norm=0.5)
    env, net, algo, buf, col = bench.build_job(a, torch.device("cuda"), 0)
    ret = 0.0
    for i in range(n):
        cs, ts = bench.one_step(a, algo, buf, col)
        if i % 500 == 0:
            ret = float(cs.returns.mean()) if len(cs.returns) else ret
            print(f"step {i}: mean episode return {ret:.3f}", flush=True)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in (net.flat.data, algo.exp_avg, algo.exp_avg_sq, buf.obs_store, buf.act_store, buf.rew_store, buf.logp_store,
              buf.vs_store, env.agent_pos):
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    print(f"steps {n} opt_step {algo.opt_step} sha256 {h.hexdigest()}")


if __name__ == "__main__":
    main()
