"""Host-boundary containers and the HBM-resident buffer / collector (mirror of tianshou.data for the path)."""
from .batch import Batch, to_numpy, to_torch, to_torch_as
from .buffer import DeviceAECReplayBuffer, DeviceVectorReplayBuffer, VectorReplayBuffer
from .collector import AsyncCollector, Collector
from .stats import CollectStats, SequenceSummaryStats

__all__ = ["Batch", "to_numpy", "to_torch", "to_torch_as", "DeviceVectorReplayBuffer", "DeviceAECReplayBuffer", "VectorReplayBuffer",
           "Collector", "AsyncCollector", "CollectStats", "SequenceSummaryStats"]
