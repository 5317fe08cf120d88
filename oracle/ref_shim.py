"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Stub shim that lets the *reference* package (pure Python, /root/reference, requires
Python 3.11 + numba/gymnasium/pettingzoo/sensai/...) import under this container's
Python 3.10 so that its own functions can be executed to generate golden vectors
(tests/golden/make_fixtures.py) and to validate the CPU restatement in oracle/.

Nothing from the reference is copied: this file only fabricates the *missing third-party
modules* (SURVEY.md section 8c lists them).  It locates the reference exclusively via
/root/reference and refuses to do anything when that path is absent (GPU box).

numba note: the reference's ``@njit`` functions (`_gae`, `_prev_index`, `_next_index`,
tianshou/algorithm/algorithm_base.py:1079, tianshou/data/buffer/manager.py:306,334) are run
un-jitted.  numba types ``gamma``/``gae_lambda`` as float64 (algorithm_base.py:343-351) so
``v_s_(f32) * gamma`` promotes to float64, whereas plain numpy>=2 keeps a python float "weak"
(result f32).  To reproduce numba's arithmetic, the fake ``njit`` converts python-float
arguments to ``np.float64`` before the call.
"""
from __future__ import annotations

import enum
import importlib
import logging as _stdlogging
import os
import sys
import types
import typing

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "tianshou"))


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def _install_py311_compat() -> None:
    import typing_extensions

    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self  # type: ignore[attr-defined]
    if not hasattr(enum, "StrEnum"):

        class StrEnum(str, enum.Enum):
            def __str__(self) -> str:
                return str(self.value)

        enum.StrEnum = StrEnum  # type: ignore[attr-defined]


def _install_numba() -> None:
    import numpy as np

    def njit(*args, **kwargs):
        def deco(fn):
            def wrapper(*a, **k):
                a = tuple(np.float64(x) if type(x) is float else x for x in a)
                k = {kk: (np.float64(v) if type(v) is float else v) for kk, v in k.items()}
                return fn(*a, **k)

            wrapper.__name__ = fn.__name__
            wrapper.__doc__ = fn.__doc__
            wrapper.py_func = fn
            return wrapper

        if len(args) == 1 and callable(args[0]) and not kwargs:
            return deco(args[0])
        return deco

    _mod("numba", njit=njit, jit=njit)


def _install_overrides() -> None:
    def override(fn=None, **_):
        if fn is None:
            return lambda f: f
        return fn

    _mod("overrides", override=override)


def _install_sensai() -> None:
    _mod("sensai")
    _mod("sensai.util")

    lg = _mod("sensai.util.logging")
    for k in dir(_stdlogging):
        if not k.startswith("__"):
            setattr(lg, k, getattr(_stdlogging, k))
    lg.set_configure_callback = lambda *a, **k: None
    lg.datetime_tag = lambda: "19700101-000000"
    lg.run_main = lambda fn, *a, **k: fn()
    lg.run_cli = lambda fn, *a, **k: fn()
    lg.configure = lambda *a, **k: None
    lg.add_file_logger = lambda *a, **k: None
    lg.FileLoggerContext = object

    def mark_used(*a, **k):
        return None

    def count_none(*args):
        return sum(1 for a in args if a is None)

    _mod("sensai.util.helper", mark_used=mark_used, count_none=count_none)

    def pickle_hash(o, *a, **k):
        import hashlib
        import pickle

        return hashlib.sha1(pickle.dumps(o)).hexdigest()

    _mod("sensai.util.hash", pickle_hash=pickle_hash)

    def setstate(cls, obj, state, new_default_properties=None, **_):
        if new_default_properties:
            for k, v in new_default_properties.items():
                state.setdefault(k, v)
        obj.__dict__ = state

    def dump_pickle(obj, path, *a, **k):
        import pickle

        with open(path, "wb") as f:
            pickle.dump(obj, f)

    def load_pickle(path, *a, **k):
        import pickle

        with open(path, "rb") as f:
            return pickle.load(f)

    _mod("sensai.util.pickle", setstate=setstate, dump_pickle=dump_pickle, load_pickle=load_pickle)

    class ToStringMixin:
        def __str__(self) -> str:
            return f"{self.__class__.__name__}"

        def _tostring_excludes(self):
            return []

        def _tostring_includes(self):
            return []

        def _tostring_additional_entries(self):
            return {}

        def _tostring_exclude_private(self):
            return False

        def pprint(self, *a, **k):
            print(str(self))

        def pprints(self, *a, **k):
            return str(self)

    _mod("sensai.util.string", ToStringMixin=ToStringMixin)

    class GitStatus:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    _mod("sensai.util.git", GitStatus=GitStatus, git_status=lambda *a, **k: GitStatus())


def _install_misc() -> None:
    class _Placeholder:
        def __init__(self, *a, **k):
            raise RuntimeError("h5py is stubbed in the oracle shim")

    _mod("h5py", File=_Placeholder, Dataset=_Placeholder, Group=_Placeholder)

    def DeepDiff(a, b, *args, **kwargs):
        import pickle

        try:
            return {} if pickle.dumps(a) == pickle.dumps(b) else {"changed": True}
        except Exception:
            return {"changed": True}

    _mod("deepdiff", DeepDiff=DeepDiff)

    _mod("tensorboard")
    _mod("tensorboard.backend")
    _mod("tensorboard.backend.event_processing")
    _mod("tensorboard.backend.event_processing.event_accumulator", EventAccumulator=object)
    _mod("tensorboard.backend.event_processing.event_file_loader", EventFileLoader=object)
    try:
        importlib.import_module("torch.utils.tensorboard")
    except Exception:
        import torch.utils  # noqa: F401

        class SummaryWriter:
            def __init__(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass

            def flush(self):
                pass

            def close(self):
                pass

        _mod("torch.utils.tensorboard", SummaryWriter=SummaryWriter)


def _install_gymnasium() -> None:
    import numpy as np

    class Space:
        def __init__(self, shape=None, dtype=None, seed=None):
            self._shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self._rng = np.random.default_rng(seed)

        @property
        def shape(self):
            return self._shape

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        @property
        def np_random(self):
            return self._rng

        def contains(self, x):
            return True

        def __contains__(self, x):
            return self.contains(x)

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            super().__init__(shape, dtype, seed)
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()

        def sample(self, mask=None):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def __eq__(self, o):
            return (
                isinstance(o, Box)
                and self.shape == o.shape
                and np.array_equal(self.low, o.low)
                and np.array_equal(self.high, o.high)
            )

        def __hash__(self):
            return hash((self.shape, self.low.tobytes(), self.high.tobytes()))

    class Discrete(Space):
        def __init__(self, n, seed=None, start=0):
            super().__init__((), np.int64, seed)
            self.n = int(n)
            self.start = int(start)

        def sample(self, mask=None):
            if mask is not None:
                valid = np.nonzero(np.asarray(mask))[0]
                return int(self.start + self._rng.choice(valid))
            return int(self.start + self._rng.integers(self.n))

        def contains(self, x):
            return self.start <= int(x) < self.start + self.n

        def __eq__(self, o):
            return isinstance(o, Discrete) and self.n == o.n and self.start == o.start

        def __hash__(self):
            return hash((self.n, self.start))

    class MultiDiscrete(Space):
        def __init__(self, nvec, dtype=np.int64, seed=None):
            self.nvec = np.asarray(nvec, dtype=dtype)
            super().__init__(self.nvec.shape, dtype, seed)

        def sample(self, mask=None):
            return (self._rng.random(self.nvec.shape) * self.nvec).astype(self.dtype)

        def __eq__(self, o):
            return isinstance(o, MultiDiscrete) and np.array_equal(self.nvec, o.nvec)

        def __hash__(self):
            return hash(self.nvec.tobytes())

    class MultiBinary(Space):
        def __init__(self, n, seed=None):
            self.n = n
            super().__init__((n,) if np.isscalar(n) else tuple(n), np.int8, seed)

        def sample(self, mask=None):
            return self._rng.integers(0, 2, size=self.shape).astype(self.dtype)

        def __eq__(self, o):
            return isinstance(o, MultiBinary) and self.shape == o.shape

        def __hash__(self):
            return hash(self.shape)

    class Dict(Space):  # noqa: A001
        def __init__(self, spaces=None, seed=None, **kw):
            super().__init__(None, None, seed)
            self.spaces = dict(spaces or {}, **kw)

        def sample(self, mask=None):
            return {k: s.sample() for k, s in self.spaces.items()}

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def items(self):
            return self.spaces.items()

        def values(self):
            return self.spaces.values()

        def __eq__(self, o):
            return isinstance(o, Dict) and self.spaces == o.spaces

        def __hash__(self):
            return hash(tuple(self.spaces))

    class Tuple(Space):  # noqa: A001
        def __init__(self, spaces, seed=None):
            super().__init__(None, None, seed)
            self.spaces = tuple(spaces)

        def sample(self, mask=None):
            return tuple(s.sample() for s in self.spaces)

        def __getitem__(self, i):
            return self.spaces[i]

        def __len__(self):
            return len(self.spaces)

        def __eq__(self, o):
            return isinstance(o, Tuple) and self.spaces == o.spaces

        def __hash__(self):
            return hash(self.spaces)

    class Env:
        metadata: dict = {}
        render_mode = None
        spec = None
        observation_space = None
        action_space = None
        _np_random = None

        def reset(self, *, seed=None, options=None):
            if seed is not None or self._np_random is None:
                self._np_random = np.random.default_rng(seed)
            return None, {}

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random = np.random.default_rng()
            return self._np_random

        @np_random.setter
        def np_random(self, v):
            self._np_random = v

        @property
        def unwrapped(self):
            return self

        def step(self, action):
            raise NotImplementedError

        def render(self):
            return None

        def close(self):
            return None

        def __class_getitem__(cls, item):
            return cls

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self.env, name)

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def reset(self, **kw):
            return self.env.reset(**kw)

        def step(self, action):
            return self.env.step(action)

        def render(self):
            return self.env.render()

        def close(self):
            return self.env.close()

        def __class_getitem__(cls, item):
            return cls

    class ObservationWrapper(Wrapper):
        def reset(self, **kw):
            obs, info = self.env.reset(**kw)
            return self.observation(obs), info

        def step(self, action):
            o, r, te, tr, i = self.env.step(action)
            return self.observation(o), r, te, tr, i

    class ActionWrapper(Wrapper):
        def step(self, action):
            return self.env.step(self.action(action))

    class RewardWrapper(Wrapper):
        pass

    def make(*a, **k):
        raise RuntimeError("gymnasium.make is not available in the oracle shim")

    g = _mod(
        "gymnasium",
        Env=Env,
        Wrapper=Wrapper,
        ObservationWrapper=ObservationWrapper,
        ActionWrapper=ActionWrapper,
        RewardWrapper=RewardWrapper,
        Space=Space,
        make=make,
        __version__="0.29.1",
    )
    sp = _mod(
        "gymnasium.spaces",
        Space=Space,
        Box=Box,
        Discrete=Discrete,
        MultiDiscrete=MultiDiscrete,
        MultiBinary=MultiBinary,
        Dict=Dict,
        Tuple=Tuple,
    )
    _mod("gymnasium.spaces.discrete", Discrete=Discrete)
    _mod("gymnasium.spaces.box", Box=Box)
    g.spaces = sp
    _mod("gymnasium.core", Env=Env, Wrapper=Wrapper, ObsType=typing.Any, ActType=typing.Any)
    _mod("gymnasium.envs")
    _mod("gymnasium.envs.registration", EnvSpec=object)
    _mod("gymnasium.wrappers")


def _install_pettingzoo() -> None:
    class AECEnv:
        pass

    class ParallelEnv:
        pass

    class BaseWrapper(AECEnv):
        def __init__(self, env):
            self.env = env

    _mod("pettingzoo", __version__="1.24.2")
    _mod("pettingzoo.utils")
    _mod("pettingzoo.utils.env", AECEnv=AECEnv, ParallelEnv=ParallelEnv)
    _mod("pettingzoo.utils.wrappers", BaseWrapper=BaseWrapper)


_INSTALLED = False


def install() -> None:
    """Install the stubs and put the reference on sys.path (idempotent)."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError(f"{REFERENCE_ROOT} not present: the reference cannot be imported here")
    _install_py311_compat()
    for name, fn in (
        ("numba", _install_numba),
        ("overrides", _install_overrides),
        ("sensai", _install_sensai),
        ("gymnasium", _install_gymnasium),
        ("pettingzoo", _install_pettingzoo),
    ):
        try:
            importlib.import_module(name)
        except Exception:
            fn()
    _install_misc()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # never write .pyc into the read-only reference
    _INSTALLED = True


def import_reference():
    """Return the imported reference ``tianshou`` package."""
    install()
    return importlib.import_module("tianshou")
