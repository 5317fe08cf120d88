"""Host vector envs: the `BaseVectorEnv` contract with the in-process (for-loop) worker.

Mirror of /root/reference/tianshou/env/venvs.py:25-386 for the synchronous case -- `len(venv)`,
`reset(env_id, **kw) -> (obs[R], info[R])`, `step(action, id) -> (obs, rew, terminated, truncated, info)`
with `info[i]["env_id"]`, `seed`, `get_env_attr/set_env_attr`, reserved gym keys forwarded to the workers,
object-dtype arrays for ragged/dict observations (venvs.py:227-232, 310-314).  Subprocess / shared-memory /
Ray workers are CPU scale-out mechanisms that the device vector env (env/mpe.py: every env stepped by one kernel)
replaces and are not built.  The ASYNC protocol of the vector env (`wait_num` / `timeout`, venvs.py:269-309: `step(action,
id)` hands actions to the envs in `id` and returns the results of whichever envs are ready -- at least `wait_num` of
them --, `step(None)` fetches unfinished calls, envs that are stepping may not be touched) is mirrored for the in-process
worker: as in the reference's `DummyEnvWorker.wait` every waiting env is ready at once, and `ready_selector` is the seam where
a worker pool with real latencies decides (tests script partial readiness through it; `data.collector.AsyncCollector` is
the consumer).
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
        self._env_fns = env_fns
        self.workers = [fn() for fn in env_fns]
        self.env_num = len(env_fns)
        # venvs.py:108-122
        self.wait_num = wait_num or len(env_fns)
        assert 1 <= self.wait_num <= len(env_fns), f"wait_num should be in [1, {len(env_fns)}], but got {wait_num}"
        self.timeout = timeout
        assert self.timeout is None or self.timeout > 0, f"timeout is {timeout}, it should be positive if provided!"
        self.is_async = self.wait_num != len(env_fns) or timeout is not None
        self.waiting_id: list[int] = []                # envs that have been handed an action and not returned yet
        self._pending: dict[int, Any] = {}             # their results (the in-process worker steps at once: DummyEnvWorker.send)
        self.ready_id = list(range(self.env_num))      # envs that may be handed an action
        # which of the waiting envs return from this step() call: positions into `waiting_id`, at least one.  Default: all of
        # them (the reference's DummyEnvWorker.wait: "sequential EnvWorker objects are always ready").
        self.ready_selector: Callable[[list[int], int], list[int]] = lambda waiting, wait_num: list(range(len(waiting)))
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

    def _assert_id(self, id) -> None:  # noqa: A002  (venvs.py:186-193)
        for i in id:
            assert i not in self.waiting_id, f"Cannot interact with environment {i} which is stepping now."
            assert i in self.ready_id, f"Can only interact with ready environments {self.ready_id}."

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
        if self.is_async:
            self._assert_id(self._wrap_id(env_id))
        ret = [self.workers[j].reset(**kwargs) for j in self._wrap_id(env_id)]
        assert isinstance(ret[0], tuple | list) and len(ret[0]) == 2 and isinstance(ret[0][1], dict), \
            "The environment does not adhere to the Gymnasium's API."
        return self._stack_obs([r[0] for r in ret]), np.array([r[1] for r in ret])

    def step(self, action, id=None):  # noqa: A002
        self._assert_is_not_closed()
        ids = self._wrap_id(id)
        result = []
        if not self.is_async:
            if action is None:
                raise ValueError("action must be not-None for non-async")
            assert len(action) == len(ids)
            for a, j in zip(action, ids, strict=True):
                env_return = self.workers[j].step(a)
                env_return[-1]["env_id"] = j
                result.append(env_return)
        else:  # venvs.py:289-309
            if action is not None:
                self._assert_id(ids)
                assert len(action) == len(ids)
                for a, j in zip(action, ids, strict=True):
                    self._pending[int(j)] = self.workers[j].step(a)
                    self.waiting_id.append(int(j))
                self.ready_id = [x for x in self.ready_id if x not in ids]
            if not self.waiting_id:
                raise RuntimeError("async step(None) with no environment stepping")
            take = list(dict.fromkeys(self.ready_selector(list(self.waiting_id), self.wait_num)))  # (results in THIS order)
            assert take and all(0 <= k < len(self.waiting_id) for k in take), "ready_selector must name waiting envs"
            returned = [self.waiting_id[k] for k in take]
            self.waiting_id = [j for k, j in enumerate(self.waiting_id) if k not in set(take)]
            for j in returned:
                env_return = self._pending.pop(j)
                env_return[-1]["env_id"] = j
                result.append(env_return)
                self.ready_id.append(j)
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
