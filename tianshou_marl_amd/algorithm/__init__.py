"""Algorithms on the device-resident rollout (mirror of tianshou.algorithm for the north-star path)."""
from .ppo import PPO, policy_within_training_step
from .ppo_generic import GenericPPO

__all__ = ["PPO", "GenericPPO", "policy_within_training_step"]
