#!/usr/bin/env python3
"""Host-side wall-clock split of one step of the configs[4] shard (512 simple_tag worlds, grouped PPO, league trainer):
collect / per-agent batch extraction / learn() of each team / the trainer's step.  Every phase is followed by a device
synchronisation, so a phase's figure is its GPU time plus its host time (run it under rocprofv3 --kernel-trace --stats for
the kernel split: that is how the 682 us single-lane GAE scan was found).

    python tools/diag_tag_step.py
"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd.algorithm.multiagent import FlexibleMultiAgentPolicyManager, LeaguePlayTrainer, agent_batches_from_buffer
from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
from tianshou_marl_amd.data.collector import Collector
from tianshou_marl_amd.env.mpe_tag import DeviceSimpleTagVectorEnv
from tianshou_marl_amd.utils.net import DiscreteActorCritic
dev="cuda"; n_env,T=512,25
env = DeviceSimpleTagVectorEnv(n_env, device=dev, seed=1, max_cycles=T); N=env.n_agent
mk = lambda s: PPO(net=DiscreteActorCritic(env.obs_dim, 5, 64, device=dev, seed=s), seed=s, lr=3e-4, shuffle="device")
teams = {"adversaries": mk(1), "good": mk(2)}
mgr = FlexibleMultiAgentPolicyManager(teams, env, mode="grouped", agent_groups=env.agent_groups)
buf = DeviceVectorReplayBuffer(n_env*T, n_env, N, env.obs_dim, device=dev)
col = Collector(mgr, env, buf); col.reset()
trainer = LeaguePlayTrainer(mgr, matchmaking="random"); np.random.seed(0)
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
for it in range(60):
    with policy_within_training_step(mgr):
        t0=time.perf_counter(); cs = col.collect(n_step=n_env*T); 
        if it>=10: tick("collect", t0)
        t0=time.perf_counter(); batch = agent_batches_from_buffer(buf, env.agents, only=["agent_0","adversary_0"])
        batch["good"], batch["adversaries"] = batch["agent_0"], batch["adversary_0"]
        if it>=10: tick("extract", t0)
        t0=time.perf_counter(); l1 = teams["good"].learn(batch["good"])
        if it>=10: tick("learn good", t0)
        t0=time.perf_counter(); l2 = teams["adversaries"].learn(batch["adversaries"])
        if it>=10: tick("learn adv", t0)
        t0=time.perf_counter(); losses = trainer.train_step(batch)
        if it>=10: tick("train_step (both again)", t0)
    col.reset_buffer(keep_statistics=True)
print({k: round(v/50*1e3,3) for k,v in acc.items()})
