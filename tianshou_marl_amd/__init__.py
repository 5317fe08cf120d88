"""tianshou_marl_amd -- MI355X-native rollout + update engine behind tianshou_marl's API surface.

Only the data-parallel hot path of the reference lives here: device
VectorReplayBuffer, GAE, categorical head, PPO clip loss, optimizer step, agent dispatch, CTDE
global state -- hand-written HIP for gfx950 behind the C-ABI of include/tsmarl.h -- plus the
host-side mirror of the reference's Collector / Batch / policy / trainer interfaces.
"""
__version__ = "0.1.0"
