"""(GPU box) where the HOST spends a step of the tag job (bench.run_tag's step: collect + agent batches + league train step + reset):
cProfile of 300 steps, statistics read one step late as the bench does.  python tools/host_profile_tag.py [n_env]"""
import cProfile, io, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tianshou_marl_amd.algorithm.multiagent import FlexibleMultiAgentPolicyManager, LeaguePlayTrainer, agent_batches_from_buffer
from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
from tianshou_marl_amd.data.collector import Collector
from tianshou_marl_amd.env.mpe_tag import DeviceSimpleTagVectorEnv
from tianshou_marl_amd.utils.net import DiscreteActorCritic
from tianshou_marl_amd.utils.host import limit_host_threads

limit_host_threads()
device = torch.device("cuda")
n_env, T = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 25
env = DeviceSimpleTagVectorEnv(n_env, device=device, seed=1626, max_cycles=T)
N = env.n_agent
mk = lambda s: PPO(net=DiscreteActorCritic(env.obs_dim, 5, 64, device=device, seed=s), seed=s, lr=3e-4, shuffle="device", async_stats=True)  # noqa: E731
teams = {"adversaries": mk(1626), "good": mk(1627)}
mgr = FlexibleMultiAgentPolicyManager(teams, env, mode="grouped", agent_groups=env.agent_groups)
buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=device)
col = Collector(mgr, env, buf, async_stats=True)
col.reset()
trainer = LeaguePlayTrainer(mgr, matchmaking="random")
np.random.seed(1626)
seg = {"collect": 0.0, "batches": 0.0, "train": 0.0, "reset": 0.0, "resolve": 0.0}


def step():
    with policy_within_training_step(mgr):
        t0 = time.perf_counter()
        cs = col.collect(n_step=n_env * T)
        t1 = time.perf_counter()
        batch = agent_batches_from_buffer(buf, env.agents, only=["agent_0", "adversary_0"], global_state=False, copies=False)
        batch["good"], batch["adversaries"] = batch["agent_0"], batch["adversary_0"]
        t2 = time.perf_counter()
        losses = trainer.train_step(batch)
        t3 = time.perf_counter()
    col.reset_buffer(keep_statistics=True)
    t4 = time.perf_counter()
    seg["collect"] += t1 - t0; seg["batches"] += t2 - t1; seg["train"] += t3 - t2; seg["reset"] += t4 - t3
    return cs, losses


def loop(n):
    prev = None
    for _ in range(n):
        cur = step()
        if prev is not None:
            t0 = time.perf_counter()
            bench._resolve(prev[0])
            for v in prev[1].values():
                float(v["loss"])
            seg["resolve"] += time.perf_counter() - t0
        prev = cur
    torch.cuda.synchronize()


loop(30)
for k in seg: seg[k] = 0.0
t0 = time.perf_counter(); loop(300); dt = time.perf_counter() - t0
print("step %.1f us; host segments us/step: %s" % (dt / 300 * 1e6, {k: round(v / 300 * 1e6, 1) for k, v in seg.items()}))
pr = cProfile.Profile(); pr.enable(); loop(300); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(sys.argv[2] if len(sys.argv) > 2 else "tottime").print_stats(60); print(s.getvalue()[:14000])
