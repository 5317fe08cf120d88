"""Algorithms on the device-resident rollout (mirror of tianshou.algorithm for the north-star path)."""
from .pg import A2C, Reinforce
from .ppo import PPO, policy_within_training_step
from .ppo_generic import GenericPPO

__all__ = ["PPO", "A2C", "Reinforce", "GenericPPO", "policy_within_training_step"]
