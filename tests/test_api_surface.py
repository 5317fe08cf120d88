"""Drop-in boundary conformance: the mirror classes accept the reference's call shapes.

`tests/golden/api_surface.json` is dumped from the imported reference by tests/golden/make_api_fixtures.py
(`inspect.signature` of every method the reference's trainer / collector / MARL trainers call, the statistics
dataclasses' fields, and the attributes `OnPolicyTrainer` touches, trainer/trainer.py:878-951, 1079-1109).  For every
entry the product's counterpart must take the same parameter names, at the same positions for positional parameters,
with the same defaults; extra parameters are allowed only with defaults.  CPU test: nothing is launched.
"""
import dataclasses
import inspect
import json
import os

import pytest

from tianshou_marl_amd.algorithm import optim as t_optim
from tianshou_marl_amd.algorithm import pg as t_pg
from tianshou_marl_amd.algorithm import ppo as t_ppo
from tianshou_marl_amd.algorithm.multiagent import ctde as t_ctde
from tianshou_marl_amd.algorithm.multiagent import flexible_policy as t_flex
from tianshou_marl_amd.algorithm.multiagent import marl as t_marl
from tianshou_marl_amd.algorithm.multiagent import training_coordinator as t_tc
from tianshou_marl_amd.data import buffer as t_buffer
from tianshou_marl_amd.data import collector as t_collector
from tianshou_marl_amd.data import stats as t_stats
from tianshou_marl_amd.env import venvs as t_venvs
from tianshou_marl_amd.trainer import OnPolicyTrainer, OnPolicyTrainerParams
from tianshou_marl_amd.utils import ref_nets as t_nets

API = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "api_surface.json")))

MIRROR = {
    "Collector.__init__": t_collector.Collector.__init__,
    "Collector.collect": t_collector.Collector.collect,
    "Collector.reset": t_collector.Collector.reset,
    "Collector.reset_env": t_collector.Collector.reset_env,
    "Collector.reset_buffer": t_collector.Collector.reset_buffer,
    "Collector.reset_stat": t_collector.Collector.reset_stat,
    "AsyncCollector.__init__": t_collector.AsyncCollector.__init__,
    "AsyncCollector.reset": t_collector.AsyncCollector.reset,
    "AsyncCollector.reset_env": t_collector.AsyncCollector.reset_env,
    "VectorReplayBuffer.__init__": t_buffer.VectorReplayBuffer.__init__,
    "ReplayBufferManager.add": t_buffer.DeviceVectorReplayBuffer.add,
    "ReplayBufferManager.sample_indices": t_buffer.DeviceVectorReplayBuffer.sample_indices,
    "ReplayBufferManager.unfinished_index": t_buffer.DeviceVectorReplayBuffer.unfinished_index,
    "ReplayBufferManager.prev": t_buffer.DeviceVectorReplayBuffer.prev,
    "ReplayBufferManager.next": t_buffer.DeviceVectorReplayBuffer.next,
    "ReplayBufferManager.reset": t_buffer.DeviceVectorReplayBuffer.reset,
    "ReplayBuffer.sample": t_buffer.DeviceVectorReplayBuffer.sample,
    "ReplayBuffer.get_buffer_indices": t_buffer.DeviceVectorReplayBuffer.get_buffer_indices,
    "ReplayBuffer.hasnull": t_buffer.DeviceVectorReplayBuffer.hasnull,
    "ReplayBuffer.isnull": t_buffer.DeviceVectorReplayBuffer.isnull,
    "ReplayBuffer.set_array_at_key": t_buffer.DeviceVectorReplayBuffer.set_array_at_key,
    "OnPolicyAlgorithm.update": t_ppo.PPO.update,
    "PPO.__init__": t_ppo.PPO.__init__,
    "A2C.__init__": t_pg.A2C.__init__,
    "Reinforce.__init__": t_pg.Reinforce.__init__,
    "DiscreteActorPolicy.__init__": t_nets.DiscreteActorPolicy.__init__,
    "Net.__init__": t_nets.Net.__init__,
    "MLP.__init__": t_nets.MLP.__init__,
    "DiscreteActor.__init__": t_nets.DiscreteActor.__init__,
    "DiscreteCritic.__init__": t_nets.DiscreteCritic.__init__,
    "ActorCritic.__init__": t_nets.ActorCritic.__init__,
    "AdamOptimizerFactory.__init__": t_optim.AdamOptimizerFactory.__init__,
    "LRSchedulerFactoryLinear.__init__": t_optim.LRSchedulerFactoryLinear.__init__,
    "BaseVectorEnv.reset": t_venvs.BaseVectorEnv.reset,
    "BaseVectorEnv.step": t_venvs.BaseVectorEnv.step,
    "DummyVectorEnv.__init__": t_venvs.DummyVectorEnv.__init__,
    "MultiAgentPolicy.forward": t_marl.MultiAgentPolicy.forward,
    "FlexibleMultiAgentPolicyManager.__init__": t_flex.FlexibleMultiAgentPolicyManager.__init__,
    "CTDEPolicy.__init__": t_ctde.CTDEPolicy.__init__,
    "CTDEPolicy.learn": t_ctde.CTDEPolicy.learn,
    "MATrainer.__init__": t_tc.MATrainer.__init__,
    "SimultaneousTrainer.train_step": t_tc.SimultaneousTrainer.train_step,
    "SequentialTrainer.train_step": t_tc.SequentialTrainer.train_step,
    "SelfPlayTrainer.__init__": t_tc.SelfPlayTrainer.__init__,
    "SelfPlayTrainer.train_step": t_tc.SelfPlayTrainer.train_step,
    "LeaguePlayTrainer.__init__": t_tc.LeaguePlayTrainer.__init__,
    "LeaguePlayTrainer.train_step": t_tc.LeaguePlayTrainer.train_step,
    "MATrainer.save_checkpoint": t_tc.MATrainer.save_checkpoint,
    "MATrainer.load_checkpoint": t_tc.MATrainer.load_checkpoint,
    "policy_within_training_step": t_ppo.policy_within_training_step.__init__,
}

POSITIONAL = ("POSITIONAL_OR_KEYWORD", "POSITIONAL_ONLY")


def _norm(v):
    return list(v) if isinstance(v, (list, tuple)) else v


@pytest.mark.parametrize("name", sorted(API["signatures"]))
def test_mirror_accepts_the_reference_call_shape(name):
    assert name in MIRROR, f"no counterpart registered for {name}"
    ours = [p for p in inspect.signature(MIRROR[name]).parameters.values() if p.name != "self"]
    by_name = {p.name: p for p in ours}
    var_kw = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in ours)
    var_pos = any(p.kind is inspect.Parameter.VAR_POSITIONAL for p in ours)
    ours_pos = [p.name for p in ours if p.kind.name in POSITIONAL]
    ref = API["signatures"][name]
    ref_pos = [p["name"] for p in ref if p["kind"] in POSITIONAL]
    for p in ref:
        if p["kind"].startswith("VAR_"):
            continue  # reference-side *args / **kwargs are sinks, nothing to accept
        if p["name"] not in by_name:
            assert var_kw or (var_pos and p["kind"] in POSITIONAL), f"{name}: parameter `{p['name']}` is not accepted"
            continue
        o = by_name[p["name"]]
        if p["kind"] in POSITIONAL:  # positional calls must land on the same parameter
            assert o.kind.name in POSITIONAL, f"{name}: `{p['name']}` must be passable positionally"
            assert ours_pos.index(p["name"]) == ref_pos.index(p["name"]), f"{name}: `{p['name']}` sits at another position"
        if "default" in p:
            assert o.default is not inspect.Parameter.empty, f"{name}: `{p['name']}` has a default in the reference"
            if p["default"] != "<callable>" and not isinstance(p["default"], str):
                assert _norm(o.default) == _norm(p["default"]), f"{name}: default of `{p['name']}`"
    names_ref = {p["name"] for p in ref}
    for o in ours:  # anything the product adds must be optional
        if o.name not in names_ref and o.kind.name in POSITIONAL + ("KEYWORD_ONLY",):
            assert o.default is not inspect.Parameter.empty, f"{name}: extra parameter `{o.name}` has no default"


@pytest.mark.parametrize("name,cls", [("CollectStats", t_stats.CollectStats), ("SequenceSummaryStats", t_stats.SequenceSummaryStats),
                                      ("TrainingStats", t_stats.TrainingStats), ("A2CTrainingStats", t_stats.A2CTrainingStats),
                                      ("OnPolicyTrainerParams", OnPolicyTrainerParams)])
def test_statistics_dataclasses_have_the_reference_fields(name, cls):
    ours = [f.name for f in dataclasses.fields(cls)]
    assert ours[:len(API["dataclass_fields"][name])] == API["dataclass_fields"][name] or \
        set(API["dataclass_fields"][name]) <= set(ours)
    assert set(API["dataclass_fields"][name]) <= set(ours)


def test_trainer_params_defaults_match():
    ref = API["default_hyperparameters"]["OnPolicyTrainerParams"]
    ours = {f.name: f.default for f in dataclasses.fields(OnPolicyTrainerParams)}
    for k, v in ref.items():
        assert ours[k] == v, k
    ref = API["default_hyperparameters"]["PPO"]
    sig = inspect.signature(t_ppo.PPO.__init__).parameters
    for k, v in ref.items():
        assert sig[k].default == v, k


def test_objects_expose_what_the_on_policy_trainer_touches():
    touches = API["trainer_touches"]
    for attr in touches["algorithm"]:
        assert hasattr(t_ppo.PPO, attr), attr
    for attr in touches["train_collector"]:
        assert hasattr(t_collector.Collector, attr) or attr in ("buffer", "collect_step", "collect_episode", "collect_time")
    src = inspect.getsource(t_collector.Collector.__init__)
    for attr in ("buffer", "collect_step", "collect_episode", "collect_time"):
        assert f"self.{attr}" in src, attr
    for attr in touches["buffer"]:
        assert hasattr(t_buffer.DeviceVectorReplayBuffer, attr), attr
    for attr in touches["collect_stats"]:
        assert attr in {f.name for f in dataclasses.fields(t_stats.CollectStats)}, attr
    for attr in touches["training_stats"]:
        assert hasattr(t_stats.TrainingStats, attr) or attr in {f.name for f in dataclasses.fields(t_stats.TrainingStats)}
    for m in API["map_training_stats_methods"]:
        assert hasattr(t_stats.MapTrainingStats, m), m
    assert hasattr(OnPolicyTrainer, "run") and hasattr(t_ppo.PPO, "run_training") and hasattr(t_ppo.PPO, "create_trainer")
