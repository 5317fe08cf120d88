"""Host-boundary containers and the HBM-resident buffer / collector (mirror of tianshou.data for the path)."""
from .batch import Batch, to_numpy, to_torch, to_torch_as
from .buffer import DeviceVectorReplayBuffer
from .collector import Collector
from .stats import CollectStats, SequenceSummaryStats

VectorReplayBuffer = DeviceVectorReplayBuffer  # the reference name (tianshou/data/buffer/vecbuf.py:15)

__all__ = ["Batch", "to_numpy", "to_torch", "to_torch_as", "DeviceVectorReplayBuffer", "VectorReplayBuffer",
           "Collector", "CollectStats", "SequenceSummaryStats"]
