#!/usr/bin/env python3
"""Learning curves of the headline job (simple_spread N=3, 1024 envs, shared PPO) for several configurations, next to an
INDEPENDENT PyTorch replica of the same algorithm -- diagnosis tool behind DESIGN.md "long-run behaviour".

    python tools/learning_curve.py [n_updates] [config ...]      configs: default gradclip retscale vclip replica replica_retscale

`engine:*` rows run the product (fused persistent rollout, captured update graph).  `replica*` rows share NOTHING with the
fused kernels except the env step kernel: the rollout is a plain python loop (torch MLP forward, torch.multinomial
sampling, `env.step_device`), V(obs) / V(obs_next) / log-probs are recomputed by torch, GAE is a float64 torch loop with
the reference's end-flag / value-mask rules (algorithm_base.py:631-717, 1079-1134), and the update is torch autograd +
torch.optim.Adam in the reference's minibatch order (per-agent dispatch, marl.py:251-268; ppo.py:164-224).  If the engine
and the replica show the same curve for the same hyper-parameters, the behaviour is the algorithm's, not a kernel's.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tianshou_marl_amd.algorithm import PPO, policy_within_training_step  # noqa: E402
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer  # noqa: E402
from tianshou_marl_amd.data.collector import Collector  # noqa: E402
from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv  # noqa: E402
from tianshou_marl_amd.utils.net import DiscreteActorCritic  # noqa: E402

DEV = "cuda"
E, N, T, MB = 1024, 3, 25, 4096
CONFIGS = {
    "default": {},                                   # the reference's PPO defaults (what bench.py runs)
    "gradclip": dict(max_grad_norm=0.5),
    "retscale": dict(return_scaling=True),
    "vclip": dict(value_clip=True, max_grad_norm=0.5),
    "retscale_gradclip": dict(return_scaling=True, max_grad_norm=0.5),
    "lr1e-4": dict(lr=1e-4),
    "lr1e-4_gradclip": dict(lr=1e-4, max_grad_norm=0.5),
    "lr1e-4_vclip_gradclip": dict(lr=1e-4, max_grad_norm=0.5, value_clip=True),
    "lr5e-5_gradclip": dict(lr=5e-5, max_grad_norm=0.5),
    "gamma095": dict(gamma=0.95),
    "gamma095_gradclip": dict(gamma=0.95, max_grad_norm=0.5),
    "gamma095_lr1e-4_gradclip": dict(gamma=0.95, lr=1e-4, max_grad_norm=0.5),
    "retscale_lr1e-4_gradclip": dict(return_scaling=True, lr=1e-4, max_grad_norm=0.5),
}


def engine_curve(n_updates: int, every: int, **kw) -> list:
    env = DeviceSimpleSpreadVectorEnv(E, N, device=DEV, seed=1626)
    net = DiscreteActorCritic(env.obs_dim, 5, 64, device=DEV, seed=1626)
    kw = dict(kw)
    algo = PPO(net=net, lr=kw.pop("lr", 3e-4), dispatch="per_agent", shuffle="device", seed=1626, **kw)
    buf = DeviceVectorReplayBuffer(E * T, E, N, env.obs_dim, device=DEV)
    col = Collector(algo, env, buf)
    col.reset()
    out = []
    for i in range(n_updates):
        with policy_within_training_step(algo):
            cs = col.collect(n_step=E * T)
            ts = algo.update(buf, MB, 1)
        col.reset_buffer(keep_statistics=True)
        if i % every == 0 or i == n_updates - 1:
            d = ts.get_loss_stats_dict()
            out.append((i, float(cs.returns.mean()), d["agent_0/vf_loss"], d["agent_0/ent_loss"]))
    return out


def _mlp(d_in, d_out, seed):
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(d_in, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, d_out)).to(DEV)
    for m in net:
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    return net


def replica_curve(n_updates: int, every: int, return_scaling=False, max_grad_norm=None, lr=3e-4, gamma=0.99, **_) -> list:
    env = DeviceSimpleSpreadVectorEnv(E, N, device=DEV, seed=1626)
    D = env.obs_dim
    actor, critic = _mlp(D, 5, 1), _mlp(D, 1, 2)
    params = list(actor.parameters()) + list(critic.parameters())
    opt = torch.optim.Adam(params, lr=lr)
    gen = torch.Generator(device=DEV).manual_seed(7)
    obs = env.reset_device().clone()
    ep_ret = torch.zeros(E, N, device=DEV, dtype=torch.float64)
    rms_mean, rms_var, rms_count = 0.0, 1.0, 0
    out = []
    for i in range(n_updates):
        O, O2, Ac, R, Te, Tr = [], [], [], [], [], []
        finished = []
        for _ in range(T):
            with torch.no_grad():
                probs = torch.softmax(actor(obs), -1).reshape(E * N, 5)
                if not bool(torch.isfinite(probs).all()):  # the run has diverged to NaN: stop (multinomial would assert)
                    out.append((i, float("nan"), float("nan"), float("nan")))
                    return out
                act = torch.multinomial(probs, 1, generator=gen).reshape(E, N)
            obs_next, rew, term, trunc, done = env.step_device(act.to(torch.int32))
            O.append(obs.clone()); O2.append(obs_next.clone()); Ac.append(act); R.append(rew.clone())  # noqa: E702
            Te.append(term.bool().clone()); Tr.append(trunc.bool().clone())  # noqa: E702
            ep_ret += rew.double()
            d = done.bool()
            if d.any():
                finished.append(ep_ret[d].clone())
                ep_ret[d] = 0
            obs = env.obs_cur.clone()
        O, O2, Ac, R, Te, Tr = (torch.stack(x) for x in (O, O2, Ac, R, Te, Tr))        # [T, E, N, ...]
        with torch.no_grad():
            v_s, v_n = critic(O).squeeze(-1).double(), critic(O2).squeeze(-1).double()
            logp_old = torch.log_softmax(actor(O), -1).gather(-1, Ac.unsqueeze(-1)).squeeze(-1)
        scale = float(np.sqrt(rms_var + 1e-8)) if return_scaling else 1.0
        v_s, v_n = v_s * scale, v_n * scale * (~Te)
        end = Te | Tr
        end[-1] = True                                                                # unfinished_index forcing
        delta = R.double() + gamma * v_n - v_s
        adv = torch.zeros_like(delta)
        g = torch.zeros(E, N, device=DEV, dtype=torch.float64)
        for t in range(T - 1, -1, -1):
            g = delta[t] + gamma * 0.95 * (~end[t]) * g
            adv[t] = g
        unnorm = adv + v_s
        ret = (unnorm / scale).float()
        if return_scaling:                                                             # a2c.py:144-146, statistics.py:97-114
            for a in range(N):  # the dispatcher preprocesses agent after agent (marl.py:208-249)
                x = unnorm[:, :, a].reshape(-1)
                bm, bv, bc = float(x.mean()), float(x.var(unbiased=False)), x.numel()
                delta_m, tot = bm - rms_mean, rms_count + bc
                m2 = rms_var * rms_count + bv * bc + delta_m ** 2 * rms_count * bc / tot
                rms_mean, rms_var, rms_count = rms_mean + delta_m * bc / tot, m2 / tot, tot
        adv = adv.float()
        vf_last = ent_last = 0.0
        for a in range(N):                                                            # marl.py:251-268
            o_a, act_a = O[:, :, a].reshape(-1, D), Ac[:, :, a].reshape(-1)
            lp_a, adv_a, ret_a = logp_old[:, :, a].reshape(-1), adv[:, :, a].reshape(-1), ret[:, :, a].reshape(-1)
            perm = torch.randperm(E * T, device=DEV, generator=gen)
            for s in range(0, E * T, MB):
                mb = perm[s:s + MB]
                lsm = torch.log_softmax(actor(o_a[mb]), -1)
                am = adv_a[mb]
                am = (am - am.mean()) / (am.std() + 1e-8)
                ratio = (lsm.gather(-1, act_a[mb].unsqueeze(-1)).squeeze(-1) - lp_a[mb]).exp()
                clip_loss = -torch.min(ratio * am, ratio.clamp(0.8, 1.2) * am).mean()
                vf = (ret_a[mb] - critic(o_a[mb]).squeeze(-1)).pow(2).mean()
                ent = -(lsm.exp() * lsm).sum(-1).mean()
                loss = clip_loss + 0.5 * vf - 0.01 * ent
                opt.zero_grad()
                loss.backward()
                if max_grad_norm:
                    torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
                opt.step()
                if a == 0:
                    vf_last, ent_last = float(vf), float(ent)
        if i % every == 0 or i == n_updates - 1:
            r = torch.cat(finished).mean().item() if finished else float("nan")
            out.append((i, r, vf_last, ent_last))
            print(f"replica update {i}: return {r:.2f} vf_loss {vf_last:.4g} entropy {ent_last:.3f}", file=sys.stderr, flush=True)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    which = sys.argv[2:] or ["default", "gradclip", "retscale", "replica", "replica_retscale"]
    every = max(1, n // 15)
    for name in which:
        if name.startswith("replica"):
            cfg = CONFIGS.get(name[len("replica_"):], {}) if "_" in name else {}
            curve = replica_curve(n, every, **cfg)
        else:
            curve = engine_curve(n, every, **CONFIGS[name])
        print(json.dumps({"config": name, "updates": n,
                          "curve_update_return_vfloss_entropy": [[i, round(r, 2), round(v, 3), round(e, 3)] for i, r, v, e in curve]}),
              flush=True)


if __name__ == "__main__":
    main()
