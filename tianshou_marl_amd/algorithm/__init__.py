"""Algorithms on the device-resident rollout (mirror of tianshou.algorithm for the north-star path)."""
from .ppo import PPO, policy_within_training_step

__all__ = ["PPO", "policy_within_training_step"]
