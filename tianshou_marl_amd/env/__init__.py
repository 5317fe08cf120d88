"""Environment wrappers and vector envs (mirror of tianshou.env for the path)."""
from .enhanced_pettingzoo_env import EnhancedPettingZooEnv
from .mpe import DeviceSimpleSpreadVectorEnv
from .mpe_tag import DeviceSimpleTagVectorEnv
from .pettingzoo_env import PettingZooEnv
from .venvs import BaseVectorEnv, DummyVectorEnv

__all__ = ["PettingZooEnv", "EnhancedPettingZooEnv", "BaseVectorEnv", "DummyVectorEnv", "DeviceSimpleSpreadVectorEnv",
           "DeviceSimpleTagVectorEnv"]
