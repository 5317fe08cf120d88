"""debug: which weights of the ctde_c3 replay differ from the reference after the N calls (dumps the arrays)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_dense as td
from tianshou_marl_amd import ops
from tianshou_marl_amd.algorithm.multiagent import *  # noqa
from tianshou_marl_amd.algorithm.multiagent.training_coordinator import _attach_global
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
from tianshou_marl_amd.data import Batch
from tianshou_marl_amd.utils.net import FlatAdam
g = np.load(os.path.join(td.GOLD, "ctde_c3.npz"))
N, D, A, H, E, T = (int(x) for x in g["dims"])
out = {}
for variant in ("chain", "early"):
    for path in ("store_eager", "copies"):
        actor = td.DecentralizedActor(D, A, hidden_dim=H, device="cuda")
        critic = td.CentralizedCritic(N * D, N, hidden_dim=H, device="cuda")
        actor.load_layers([(g[f"{variant}_init_actor_w{i}"], g[f"{variant}_init_actor_b{i}"]) for i in range(3)])
        critic.load_layers([(g[f"{variant}_init_critic_w{i}"], g[f"{variant}_init_critic_b{i}"]) for i in range(3)])
        pol = td.CTDEPolicy(actor=actor, critic=critic, optim_actor=FlatAdam(actor, lr=1e-3), optim_critic=FlatAdam(critic, lr=1e-3),
                            discount_factor=float(g["gamma"]), fused=path != "copies", graph=False)
        buf = DeviceVectorReplayBuffer(E * T, E, N, D, device="cuda")
        f = lambda k, t: g[f"{variant}_{k}"][t]
        for t in range(T):
            buf.add(Batch(obs=f("obs", t), act=f("act", t), rew=f("rew", t), terminated=f("term", t), truncated=f("trunc", t), obs_next=f("obs_next", t)))
        buf.mark_rows_chained("empty", True)
        agents = [f"agent_{i}" for i in range(N)]
        batches = td.agent_batches_from_buffer(buf, agents, copies=path == "copies")
        for a, name in enumerate(agents):
            pol.learn(_attach_global(batches, batches[name]))
            out[f"{variant}_{path}_critic_after{a+1}"] = critic.flat.data.cpu().numpy().copy()
        ref = td._wide_flat(g, f"{variant}_afterN", "critic")
        got = critic.flat.data.double().cpu().numpy()
        d = np.abs(got - ref)
        bad = np.nonzero(d > 5e-6 + 1e-5 * np.abs(ref))[0]
        print(variant, path, "bad:", bad, d[bad], "ref", ref[bad], "exp_avg_sq", pol.optim_critic.exp_avg_sq.cpu().numpy()[bad],
              "exp_avg", pol.optim_critic.exp_avg.cpu().numpy()[bad])
        out[f"{variant}_{path}_v"] = pol.optim_critic.exp_avg_sq.cpu().numpy()
np.savez_compressed("gpurun_out/ctde_c3_diff.npz", **out)
