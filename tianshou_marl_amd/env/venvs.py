"""Host vector envs: the `BaseVectorEnv` contract with the in-process (for-loop) worker.

Mirror of /root/reference/tianshou/env/venvs.py:25-386 for the synchronous case -- `len(venv)`,
`reset(env_id, **kw) -> (obs[R], info[R])`, `step(action, id) -> (obs, rew, terminated, truncated, info)`
with `info[i]["env_id"]`, `seed`, `get_env_attr/set_env_attr`, reserved gym keys forwarded to the workers,
object-dtype arrays for ragged/dict observations (venvs.py:227-232, 310-314).  Subprocess / shared-memory /
Ray workers and the async `wait_num`/`timeout` mode are CPU scale-out mechanisms that the device vector env
(env/mpe.py: every env stepped by one kernel) replaces; asking for them raises.
"""
from __future__ import annotations

from collections.abc import Callable, Sequence
from typing import Any

import numpy as np

# attributes looked up on the wrapped envs rather than on the vector env (venvs.py:13-22)
GYM_RESERVED_KEYS = ("metadata", "reward_range", "spec", "action_space", "observation_space")


class BaseVectorEnv:
    is_async = False

    def __init__(self, env_fns: Sequence[Callable[[], Any]], wait_num: int | None = None,
                 timeout: float | None = None) -> None:
        if (wait_num is not None and wait_num != len(env_fns)) or timeout is not None:
            raise NotImplementedError(
                "async vector envs (wait_num/timeout) are not part of this build: use the device vector env")
        self._env_fns = env_fns
        self.workers = [fn() for fn in env_fns]
        self.env_num = len(env_fns)
        self.wait_num = self.env_num
        self.timeout = None
        self.ready_id = list(range(self.env_num))
        self.is_closed = False

    def _assert_is_not_closed(self) -> None:
        assert not self.is_closed, f"Methods of {self.__class__.__name__} cannot be called after close."

    def __len__(self) -> int:
        return self.env_num

    def __getattribute__(self, key: str) -> Any:
        if key in GYM_RESERVED_KEYS:
            return self.get_env_attr(key)
        return super().__getattribute__(key)

    def _wrap_id(self, id=None):  # noqa: A002
        if id is None:
            return list(range(self.env_num))
        return [id] if np.isscalar(id) else id

    def get_env_attr(self, key: str, id=None) -> list:  # noqa: A002
        self._assert_is_not_closed()
        return [getattr(self.workers[j], key) for j in self._wrap_id(id)]

    def set_env_attr(self, key: str, value: Any, id=None) -> None:  # noqa: A002
        self._assert_is_not_closed()
        for j in self._wrap_id(id):
            setattr(self.workers[j], key, value)

    @staticmethod
    def _stack_obs(obs_list: list) -> np.ndarray:
        if isinstance(obs_list[0], tuple):
            raise TypeError("Tuple observation space is not supported. ", "Please change it to array or dict space")
        try:
            return np.stack(obs_list)
        except ValueError:  # ragged observations
            out = np.empty(len(obs_list), dtype=object)
            out[:] = obs_list
            return out

    def reset(self, env_id=None, **kwargs: Any) -> tuple[np.ndarray, np.ndarray]:
        self._assert_is_not_closed()
        ret = [self.workers[j].reset(**kwargs) for j in self._wrap_id(env_id)]
        assert isinstance(ret[0], tuple | list) and len(ret[0]) == 2 and isinstance(ret[0][1], dict), \
            "The environment does not adhere to the Gymnasium's API."
        return self._stack_obs([r[0] for r in ret]), np.array([r[1] for r in ret])

    def step(self, action, id=None):  # noqa: A002
        self._assert_is_not_closed()
        ids = self._wrap_id(id)
        if action is None:
            raise ValueError("action must be not-None for non-async")
        assert len(action) == len(ids)
        result = []
        for a, j in zip(action, ids, strict=True):
            env_return = self.workers[j].step(a)
            env_return[-1]["env_id"] = j
            result.append(env_return)
        obs_list, rew_list, term_list, trunc_list, info_list = tuple(zip(*result, strict=True))
        return (self._stack_obs(list(obs_list)), np.stack(rew_list), np.stack(term_list), np.stack(trunc_list),
                np.stack(info_list))

    def seed(self, seed: int | list[int] | None = None) -> list:
        self._assert_is_not_closed()
        if seed is None:
            seeds = [None] * self.env_num
        elif isinstance(seed, int):
            seeds = [seed + i for i in range(self.env_num)]
        else:
            seeds = seed
        out = []
        for w, s in zip(self.workers, seeds, strict=True):
            out.append(w.seed(s) if hasattr(w, "seed") else w.reset(seed=s))
        return out

    def render(self, **kwargs: Any) -> list:
        self._assert_is_not_closed()
        return [w.render(**kwargs) for w in self.workers]

    def close(self) -> None:
        self._assert_is_not_closed()
        for w in self.workers:
            w.close()
        self.is_closed = True


class DummyVectorEnv(BaseVectorEnv):
    """For-loop vector env (venvs.py:365-386): BASELINE configs[0] runs on this."""
