#!/usr/bin/env python3
"""Generate tests/golden/api_surface.json and tests/golden/checkpoint.npz by IMPORTING THE REFERENCE
(through oracle/ref_shim.py).  Test infrastructure; build container only (needs /root/reference).

  api_surface.json   call shapes of the drop-in boundary: `inspect.signature` of the methods the
                     reference's trainer / collector / MARL trainers call on policy, buffer, collector and algorithm
                     objects, the dataclass fields of the statistics objects, and the attributes
                     `OnPolicyTrainer._training_step/_collect_training_data/_update_step` touch
                     (trainer/trainer.py:878-951, 1079-1109).
  checkpoint.npz     `PPO.state_dict()` of the reference after the `mb64` update of ppo_update.npz (every tensor, the
                     torch-Adam `_optimizers` entry flattened), `CTDEPolicy.state_dict()` keys / shapes and the
                     `MATrainer.save_checkpoint` layout (algorithm_base.py:521-541, training_coordinator.py:225-263).
"""
from __future__ import annotations

import dataclasses
import inspect
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402
from tianshou.algorithm.algorithm_base import Algorithm, OnPolicyAlgorithm, TrainingStats  # noqa: E402
from tianshou.algorithm.modelfree.a2c import A2C, A2CTrainingStats  # noqa: E402
from tianshou.algorithm.modelfree.ppo import PPO  # noqa: E402
from tianshou.algorithm.modelfree.reinforce import DiscreteActorPolicy, Reinforce  # noqa: E402
from tianshou.algorithm.multiagent import training_coordinator as tc  # noqa: E402
from tianshou.algorithm.multiagent.ctde import CentralizedCritic, CTDEPolicy, DecentralizedActor  # noqa: E402
from tianshou.algorithm.multiagent.flexible_policy import FlexibleMultiAgentPolicyManager  # noqa: E402
from tianshou.algorithm.multiagent.marl import MapTrainingStats, MultiAgentPolicy  # noqa: E402
from tianshou.algorithm.optim import AdamOptimizerFactory, LRSchedulerFactoryLinear  # noqa: E402
from tianshou.data import AsyncCollector, Batch, Collector, CollectStats, ReplayBuffer, ReplayBufferManager, VectorReplayBuffer  # noqa: E402
from tianshou.data.stats import SequenceSummaryStats  # noqa: E402
from tianshou.env.venvs import BaseVectorEnv, DummyVectorEnv  # noqa: E402
from tianshou.trainer.trainer import OnPolicyTrainerParams  # noqa: E402
from tianshou.utils.net.common import MLP, ActorCritic, Net  # noqa: E402
from tianshou.utils.net.discrete import DiscreteActor, DiscreteCritic  # noqa: E402
from tianshou.utils.torch_utils import policy_within_training_step  # noqa: E402


def sig(fn) -> list[dict]:
    out = []
    for p in inspect.signature(fn).parameters.values():
        if p.name == "self":
            continue
        d = {"name": p.name, "kind": p.kind.name}
        if p.default is not inspect.Parameter.empty:
            try:
                json.dumps(p.default)
                d["default"] = p.default
            except TypeError:
                d["default"] = repr(p.default) if not callable(p.default) else "<callable>"
        out.append(d)
    return out


def fields(cls) -> list[str]:
    return [f.name for f in dataclasses.fields(cls)]


def make_api_surface() -> None:
    api = {
        "signatures": {
            "Collector.__init__": sig(Collector.__init__),
            "Collector.collect": sig(Collector.collect),
            "Collector.reset": sig(Collector.reset),
            "Collector.reset_env": sig(Collector.reset_env),
            "Collector.reset_buffer": sig(Collector.reset_buffer),
            "Collector.reset_stat": sig(Collector.reset_stat),
            "AsyncCollector.__init__": sig(AsyncCollector.__init__),
            "AsyncCollector.reset": sig(AsyncCollector.reset),
            "AsyncCollector.reset_env": sig(AsyncCollector.reset_env),
            "VectorReplayBuffer.__init__": sig(VectorReplayBuffer.__init__),
            "ReplayBufferManager.add": sig(ReplayBufferManager.add),
            "ReplayBufferManager.sample_indices": sig(ReplayBufferManager.sample_indices),
            "ReplayBufferManager.unfinished_index": sig(ReplayBufferManager.unfinished_index),
            "ReplayBufferManager.prev": sig(ReplayBufferManager.prev),
            "ReplayBufferManager.next": sig(ReplayBufferManager.next),
            "ReplayBufferManager.reset": sig(ReplayBufferManager.reset),
            "ReplayBuffer.sample": sig(ReplayBuffer.sample),
            "ReplayBuffer.get_buffer_indices": sig(ReplayBuffer.get_buffer_indices),
            "ReplayBuffer.hasnull": sig(ReplayBuffer.hasnull),
            "ReplayBuffer.isnull": sig(ReplayBuffer.isnull),
            "ReplayBuffer.set_array_at_key": sig(ReplayBuffer.set_array_at_key),
            "OnPolicyAlgorithm.update": sig(OnPolicyAlgorithm.update),
            "PPO.__init__": sig(PPO.__init__),
            "A2C.__init__": sig(A2C.__init__),
            "Reinforce.__init__": sig(Reinforce.__init__),
            "DiscreteActorPolicy.__init__": sig(DiscreteActorPolicy.__init__),
            "Net.__init__": sig(Net.__init__),
            "MLP.__init__": sig(MLP.__init__),
            "DiscreteActor.__init__": sig(DiscreteActor.__init__),
            "DiscreteCritic.__init__": sig(DiscreteCritic.__init__),
            "ActorCritic.__init__": sig(ActorCritic.__init__),
            "AdamOptimizerFactory.__init__": sig(AdamOptimizerFactory.__init__),
            "LRSchedulerFactoryLinear.__init__": sig(LRSchedulerFactoryLinear.__init__),
            "BaseVectorEnv.reset": sig(BaseVectorEnv.reset),
            "BaseVectorEnv.step": sig(BaseVectorEnv.step),
            "DummyVectorEnv.__init__": sig(DummyVectorEnv.__init__),
            "MultiAgentPolicy.forward": sig(MultiAgentPolicy.forward),
            "FlexibleMultiAgentPolicyManager.__init__": sig(FlexibleMultiAgentPolicyManager.__init__),
            "CTDEPolicy.__init__": sig(CTDEPolicy.__init__),
            "CTDEPolicy.learn": sig(CTDEPolicy.learn),
            "MATrainer.__init__": sig(tc.MATrainer.__init__),
            "SimultaneousTrainer.train_step": sig(tc.SimultaneousTrainer.train_step),
            "SequentialTrainer.train_step": sig(tc.SequentialTrainer.train_step),
            "SelfPlayTrainer.__init__": sig(tc.SelfPlayTrainer.__init__),
            "SelfPlayTrainer.train_step": sig(tc.SelfPlayTrainer.train_step),
            "LeaguePlayTrainer.__init__": sig(tc.LeaguePlayTrainer.__init__),
            "LeaguePlayTrainer.train_step": sig(tc.LeaguePlayTrainer.train_step),
            "MATrainer.save_checkpoint": sig(tc.MATrainer.save_checkpoint),
            "MATrainer.load_checkpoint": sig(tc.MATrainer.load_checkpoint),
            "policy_within_training_step": sig(inspect.unwrap(policy_within_training_step)),
        },
        "dataclass_fields": {
            "CollectStats": fields(CollectStats),
            "SequenceSummaryStats": fields(SequenceSummaryStats),
            "TrainingStats": fields(TrainingStats),
            "A2CTrainingStats": fields(A2CTrainingStats),
            "OnPolicyTrainerParams": fields(OnPolicyTrainerParams),
        },
        # what OnPolicyTrainer reads / calls on the objects it is handed (trainer.py:878-951, 1079-1109); each is checked
        # to exist on the reference class below
        "trainer_touches": {
            "algorithm": ["policy", "update", "state_dict", "load_state_dict", "train", "eval"],
            "policy": ["is_within_training_step"],
            "train_collector": ["collect", "buffer", "reset_buffer", "reset", "reset_env", "reset_stat", "collect_step",
                                "collect_episode", "collect_time"],
            "buffer": ["hasnull", "__len__", "reset", "sample", "unfinished_index"],
            "collect_stats": ["n_collected_steps", "n_collected_episodes", "returns", "returns_stat", "lens", "lens_stat",
                              "collect_time", "collect_speed"],
            "training_stats": ["train_time", "smoothed_loss", "get_loss_stats_dict"],
        },
        "map_training_stats_methods": [n for n in dir(MapTrainingStats) if not n.startswith("_")],
        "default_hyperparameters": {
            "PPO": {p.name: p.default for p in inspect.signature(PPO.__init__).parameters.values()
                    if isinstance(p.default, (int, float, bool, type(None)))},
            "OnPolicyTrainerParams": {f.name: f.default for f in dataclasses.fields(OnPolicyTrainerParams)
                                      if isinstance(f.default, (int, float, bool, type(None)))},
        },
    }
    # verify the hand-listed attributes against the reference classes
    probe = {"algorithm": OnPolicyAlgorithm, "train_collector": Collector, "buffer": VectorReplayBuffer, "collect_stats": CollectStats,
             "training_stats": TrainingStats}
    for who, cls in probe.items():
        for attr in api["trainer_touches"][who]:
            ok = hasattr(cls, attr) or attr in getattr(cls, "__dataclass_fields__", {}) or \
                attr in ("policy", "buffer", "collect_step", "collect_episode", "collect_time")  # instance attributes set in __init__
            assert ok, (who, attr)
    path = os.path.join(HERE, "api_surface.json")
    with open(path, "w") as f:
        json.dump(api, f, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")


def make_checkpoint() -> None:
    import make_fixtures as mf

    out = {}
    rng = np.random.default_rng(7)
    algo, actor, critic = mf.build_ppo(18, 5, [64, 64], seed=3)
    buf = mf.fill_vector_buffer(rng, 8, 25, 18, p_term=0.03, trunc_at=25, n_act=5)
    batch, indices = buf.sample(0)
    with policy_within_training_step(algo.policy):
        pb = algo._preprocess_batch(batch, buf, indices)
        np.random.seed(11)
        with torch.enable_grad():
            algo.train()
            algo._update_with_batch(pb, 64, 2)
    sd = algo.state_dict()
    keys = [k for k in sd if k != "_optimizers"]
    out["ppo_keys"] = np.array(keys)
    for k in keys:
        out["ppo/" + k] = sd[k].detach().numpy().copy()
    opt = sd["_optimizers"]
    assert len(opt) == 1 and set(opt[0]) == {"state", "param_groups"}
    pg = opt[0]["param_groups"][0]
    out["ppo_opt_param_group_keys"] = np.array(sorted(pg))
    out["ppo_opt_params"] = np.array(pg["params"])
    out["ppo_opt_hyper"] = np.array([pg["lr"], pg["betas"][0], pg["betas"][1], pg["eps"], pg["weight_decay"]], np.float64)
    for i, st in opt[0]["state"].items():
        assert set(st) == {"step", "exp_avg", "exp_avg_sq"}
        out[f"ppo_opt/{i}/step"] = np.asarray(st["step"].item(), np.float64)
        out[f"ppo_opt/{i}/exp_avg"] = st["exp_avg"].numpy().copy()
        out[f"ppo_opt/{i}/exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
    out["ppo_opt_n_state"] = len(opt[0]["state"])
    # the state the reference's load_state_dict expects back: round trip inside the reference itself
    algo2, _, _ = mf.build_ppo(18, 5, [64, 64], seed=99)
    algo2.load_state_dict(algo.state_dict())
    for (k, a), (_, b) in zip(algo.state_dict().items(), algo2.state_dict().items()):
        if k != "_optimizers":
            assert torch.equal(a, b)
    # CTDE policy: actor / critic modules with fc1..fc3
    torch.manual_seed(0)
    act = DecentralizedActor(obs_dim=6, action_dim=3, hidden_dim=16)
    cri = CentralizedCritic(global_obs_dim=12, n_agents=2, hidden_dim=16)
    pol = CTDEPolicy(actor=act, critic=cri, optim_actor=torch.optim.Adam(act.parameters(), lr=1e-3),
                     optim_critic=torch.optim.Adam(cri.parameters(), lr=1e-3),
                     observation_space=mf.gym.spaces.Box(-np.inf, np.inf, (6,)), action_space=mf.gym.spaces.Discrete(3))
    csd = pol.state_dict()
    out["ctde_keys"] = np.array(list(csd))
    out["ctde_shapes"] = np.array([json.dumps(list(v.shape)) for v in csd.values()])
    # MATrainer.save_checkpoint layout (training_coordinator.py:225-263)
    class _Mgr:
        policies = {"agent_0": pol}
        mode = "independent"
        agents = ["agent_0"]

    tr = tc.SimultaneousTrainer(_Mgr())
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ck.pt")
        tr.save_checkpoint(path)
        ck = torch.load(path, weights_only=False)
    out["matrainer_ckpt_keys"] = np.array(sorted(ck))
    out["matrainer_trainer_state_keys"] = np.array(sorted(ck["trainer_state"]))
    out["matrainer_policy_ids"] = np.array(sorted(ck["policies"]))
    path = os.path.join(HERE, "checkpoint.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


def make_batch_ops() -> None:
    """The reference's `Batch` run through tests/golden/batch_cases.py -> batch_ops.npz (row a18)."""
    import batch_cases

    out = {}
    for name, res in batch_cases.cases(Batch).items():
        for path, arr in batch_cases.flatten(res).items():
            out[f"{name}::{path}"] = arr
    path = os.path.join(HERE, "batch_ops.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    which = sys.argv[1:] or ["api_surface", "checkpoint", "batch_ops"]
    for w in which:
        globals()["make_" + w]()
