"""Multi-agent layer: policy containers, dispatch, CTDE and the training coordinators
(mirror of /root/reference/tianshou/algorithm/multiagent/__init__.py)."""
from ...data.stats import MapTrainingStats
from .ctde import CentralizedCritic, CTDEPolicy, DecentralizedActor, GlobalStateConstructor
from .flexible_policy import FlexibleMultiAgentPolicyManager
from .marl import MARLDispatcher, MultiAgentOnPolicyAlgorithm, MultiAgentPolicy
from .training_coordinator import (
    LeaguePlayTrainer,
    MATrainer,
    SelfPlayTrainer,
    SequentialTrainer,
    SimultaneousTrainer,
    agent_batches_from_buffer,
)

__all__ = [
    "MultiAgentPolicy", "MultiAgentOnPolicyAlgorithm", "MARLDispatcher", "MapTrainingStats",
    "FlexibleMultiAgentPolicyManager", "MATrainer", "SimultaneousTrainer", "SequentialTrainer",
    "SelfPlayTrainer", "LeaguePlayTrainer", "agent_batches_from_buffer", "CTDEPolicy", "GlobalStateConstructor",
    "DecentralizedActor", "CentralizedCritic",
]
