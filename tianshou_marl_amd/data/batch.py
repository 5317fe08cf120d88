"""Batch -- the dict-of-arrays container at the host boundary.

Mirrors the part of `tianshou.data.Batch` that the hot path touches
(/root/reference/tianshou/data/batch.py:632-1385): attribute + key access, numpy-style
indexing of every leaf (`__getitem__` :723-747, `__setitem__` :778-802), `cat` / `stack`
(:1045-1133), `split` incl. the merge_last rule (:1209-1225), `to_numpy` / `to_torch`,
`get_keys`, `is_empty`, `hasnull` / `isnull`.  Leaves are numpy arrays, torch tensors (host or HBM)
or nested Batches.  This is host plumbing: the device path keeps a fixed SoA schema and only
materialises a Batch when a caller asks for one.
"""
from __future__ import annotations

from collections.abc import Iterator, Sequence
from numbers import Number
from typing import Any

import numpy as np
import torch

IndexType = Any


def _is_scalar(v) -> bool:
    return isinstance(v, (Number, np.number, np.bool_)) or (isinstance(v, (np.ndarray, torch.Tensor)) and v.ndim == 0)


def _parse_value(v):
    """Normalise a leaf: dict -> Batch, list/tuple -> array (or stacked Batch), scalars -> np scalars."""
    if isinstance(v, Batch):
        return v
    if isinstance(v, dict):
        return Batch(v)
    if isinstance(v, np.ndarray) and v.dtype == object and v.ndim == 1 and len(v) > 0 and all(
            isinstance(e, (dict, Batch)) for e in v):
        # an object array of per-env dicts (what a vector env returns for dict observations / infos,
        # venvs.py:227-232) becomes one Batch of stacked leaves, as in the reference (batch.py:300-318)
        return Batch.stack([e if isinstance(e, Batch) else Batch(e) for e in v])
    if isinstance(v, (np.ndarray, torch.Tensor)):
        return v
    if v is None:
        return v
    if isinstance(v, (list, tuple)) and len(v) > 0 and all(isinstance(e, (dict, Batch)) for e in v):
        return Batch.stack([e if isinstance(e, Batch) else Batch(e) for e in v])
    if isinstance(v, (list, tuple)) and len(v) > 0 and all(isinstance(e, torch.Tensor) for e in v):
        return torch.stack(list(v))
    try:
        arr = np.asanyarray(v)
    except ValueError:
        arr = np.array(v, dtype=object)
    if not issubclass(arr.dtype.type, (np.bool_, np.number)):
        arr = arr.astype(object)
    return arr


class Batch:
    """Nested container of equally-long arrays with numpy-style indexing."""

    def __init__(self, batch_dict: dict | Batch | Sequence | None = None, copy: bool = False, **kwargs: Any) -> None:
        if copy:
            import copy as _copy

            batch_dict = _copy.deepcopy(batch_dict)
        if batch_dict is not None:
            if isinstance(batch_dict, (dict, Batch)):
                for k, v in batch_dict.items():
                    self.__dict__[k] = _parse_value(v)
            elif isinstance(batch_dict, (list, tuple, np.ndarray)) and len(batch_dict) > 0:
                self.stack_([b if isinstance(b, Batch) else Batch(b) for b in batch_dict])
        for k, v in kwargs.items():
            self.__dict__[k] = _parse_value(v)

    # ---- mapping protocol -------------------------------------------------------------------
    def __setattr__(self, key: str, value: Any) -> None:
        self.__dict__[key] = _parse_value(value)

    def __getattr__(self, key: str) -> Any:
        # only reached when normal lookup fails
        raise AttributeError(key)

    def __contains__(self, key: str) -> bool:
        return key in self.__dict__

    def get_keys(self):
        return self.__dict__.keys()

    keys = get_keys

    def values(self):
        return self.__dict__.values()

    def items(self):
        return self.__dict__.items()

    def get(self, k: str, d: Any = None) -> Any:
        return self.__dict__.get(k, d)

    def pop(self, k: str, *d: Any) -> Any:
        return self.__dict__.pop(k, *d)

    def to_dict(self, recursive: bool = True) -> dict:
        return {k: (v.to_dict() if recursive and isinstance(v, Batch) else v) for k, v in self.items()}

    def __getstate__(self) -> dict:
        return self.__dict__.copy()

    def __setstate__(self, state: dict) -> None:
        self.__dict__.update(state)

    # ---- indexing ---------------------------------------------------------------------------
    def __getitem__(self, index: str | IndexType) -> Any:
        if isinstance(index, str):
            return self.__dict__[index]
        if len(self.__dict__) == 0:
            raise IndexError("Cannot access item from empty Batch object.")
        out = Batch()
        for k, v in self.items():
            if isinstance(v, Batch) and len(v.__dict__) == 0:
                out.__dict__[k] = Batch()
            else:
                out.__dict__[k] = v[index]
        return out

    def __setitem__(self, index: str | IndexType, value: Any) -> None:
        if isinstance(index, str):
            self.__dict__[index] = _parse_value(value)
            return
        if not isinstance(value, (Batch, dict)):
            raise ValueError("Batch does not support non-Batch value assignment by index")
        value = value if isinstance(value, Batch) else Batch(value)
        if not set(value.get_keys()).issubset(self.get_keys()):
            raise ValueError("Creating keys is not supported by item assignment.")
        for k, v in self.items():
            if k in value:
                if isinstance(v, Batch):
                    if len(v.__dict__):
                        v[index] = value[k]
                else:
                    v[index] = value[k]
            elif isinstance(v, Batch):
                if len(v.__dict__):  # missing nested key: every leaf below is reset at the index
                    v[index] = Batch()
            else:  # missing key: reset to the "zero" of that leaf (batch.py:795-802)
                v[index] = None if (isinstance(v, np.ndarray) and v.dtype == object) else 0

    def __len__(self) -> int:
        lens = []
        for v in self.values():
            if isinstance(v, Batch):
                if len(v.__dict__) == 0:
                    continue
                lens.append(len(v))
            elif hasattr(v, "__len__") and (not isinstance(v, (np.ndarray, torch.Tensor)) or v.ndim > 0):
                lens.append(len(v))
            else:
                raise TypeError(f"Object {v} in {self} has no len()")
        if not lens:
            return 0
        return min(lens)

    def __iter__(self) -> Iterator[Batch]:
        for i in range(len(self)):
            yield self[i]

    @property
    def shape(self) -> list[int]:
        if len(self.__dict__) == 0:
            return []
        shapes = []
        for v in self.values():
            try:
                shapes.append(list(v.shape))
            except AttributeError:
                shapes.append([])
        return list(map(min, zip(*shapes))) if len(shapes) > 1 else shapes[0]

    def is_empty(self, recurse: bool = False) -> bool:
        if len(self.__dict__) == 0:
            return True
        if not recurse:
            return False
        return all(isinstance(v, Batch) and v.is_empty(True) for v in self.values())

    def __repr__(self) -> str:
        inner = ", ".join(f"{k}: {v!r}" for k, v in self.items())
        return f"{self.__class__.__name__}({inner})"

    def __eq__(self, other: Any) -> bool:
        if not isinstance(other, Batch) or set(self.get_keys()) != set(other.get_keys()):
            return False
        for k, v in self.items():
            o = other[k]
            if isinstance(v, Batch):
                if v != o:
                    return False
            elif isinstance(v, torch.Tensor):
                if not (isinstance(o, torch.Tensor) and torch.equal(v, o)):
                    return False
            elif not np.array_equal(np.asarray(v), np.asarray(o)):
                return False
        return True

    __hash__ = None  # type: ignore[assignment]

    # ---- conversion -------------------------------------------------------------------------
    def to_numpy(self) -> Batch:
        out = Batch()
        for k, v in self.items():
            if isinstance(v, Batch):
                out.__dict__[k] = v.to_numpy()
            elif isinstance(v, torch.Tensor):
                out.__dict__[k] = v.detach().cpu().numpy()
            else:
                out.__dict__[k] = v
        return out

    def to_numpy_(self) -> None:
        self.__dict__.update(self.to_numpy().__dict__)

    def to_torch(self, dtype: torch.dtype | None = None, device: str | torch.device = "cpu") -> Batch:
        out = Batch()
        for k, v in self.items():
            if isinstance(v, Batch):
                out.__dict__[k] = v.to_torch(dtype, device)
            elif isinstance(v, torch.Tensor):
                out.__dict__[k] = v.to(device=device, dtype=dtype or v.dtype)
            elif isinstance(v, np.ndarray) and v.dtype != object:
                # as the reference (batch.py:899-901): a numpy leaf keeps ITS dtype, `dtype` converts torch leaves only
                out.__dict__[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device)
            else:
                out.__dict__[k] = v
        return out

    def to_torch_(self, dtype: torch.dtype | None = None, device: str | torch.device = "cpu") -> None:
        self.__dict__.update(self.to_torch(dtype, device).__dict__)

    # ---- cat / stack ------------------------------------------------------------------------
    @staticmethod
    def _cat_leaves(vals: list, axis_fn_np, axis_fn_t):
        if all(isinstance(v, torch.Tensor) for v in vals):
            return axis_fn_t(vals)
        return axis_fn_np([v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v) for v in vals])

    def cat_(self, batches: Batch | Sequence[dict | Batch]) -> None:
        if isinstance(batches, (Batch, dict)):
            batches = [batches]
        batches = [b if isinstance(b, Batch) else Batch(b) for b in batches]
        batches = [b for b in batches if len(b.__dict__) > 0]
        if len(self.__dict__) > 0:
            batches = [Batch(dict(self.__dict__)), *batches]
        if not batches:
            return
        keys = list(batches[0].get_keys())
        for b in batches[1:]:
            if set(b.get_keys()) != set(keys):
                raise ValueError("Batch.cat_ requires identical keys in every batch")
        for k in keys:
            vals = [b[k] for b in batches]
            if all(isinstance(v, Batch) for v in vals):
                nb = Batch()
                nb.cat_(vals)
                self.__dict__[k] = nb
            else:
                self.__dict__[k] = self._cat_leaves(vals, np.concatenate, lambda t: torch.cat(t, dim=0))

    @staticmethod
    def cat(batches: Sequence[dict | Batch]) -> Batch:
        out = Batch()
        out.cat_(batches)
        return out

    def stack_(self, batches: Sequence[dict | Batch], axis: int = 0) -> None:
        batches = [b if isinstance(b, Batch) else Batch(b) for b in batches]
        batches = [b for b in batches if len(b.__dict__) > 0]
        if not batches:
            return
        if len(self.__dict__) > 0:
            batches = [Batch(dict(self.__dict__)), *batches]
        # keys are visited in the order of the FIRST batch (never set order, quirk Q5)
        keys = list(batches[0].get_keys())
        for b in batches[1:]:  # partially shared keys: appended in first-appearance order
            keys += [k for k in b.get_keys() if k not in keys]
        for k in keys:
            # a batch that lacks the key contributes zeros of the shape the others have (batch.py:1103-1133:
            # "batches that do not have these keys will be padded by zeros with appropriate shapes")
            template = next(b[k] for b in batches if k in b)
            vals = [b[k] if k in b else _zeros_like_value(template) for b in batches]
            if all(isinstance(v, Batch) for v in vals):
                nb = Batch()
                nb.stack_(vals, axis)
                self.__dict__[k] = nb
            else:
                self.__dict__[k] = self._cat_leaves(
                    vals, lambda a: _stack_np(a, axis), lambda t: torch.stack(t, dim=axis))

    @staticmethod
    def stack(batches: Sequence[dict | Batch], axis: int = 0) -> Batch:
        out = Batch()
        out.stack_(batches, axis)
        return out

    def update(self, batch: dict | Batch | None = None, **kwargs: Any) -> None:
        if batch is not None:
            for k, v in (batch.items() if isinstance(batch, (dict, Batch)) else []):
                self.__dict__[k] = _parse_value(v)
        for k, v in kwargs.items():
            self.__dict__[k] = _parse_value(v)

    def empty_(self, index: IndexType | None = None) -> Batch:
        for v in self.values():
            if isinstance(v, Batch):
                v.empty_(index)
            elif isinstance(v, torch.Tensor):
                if index is None:
                    v.zero_()
                else:
                    v[index] = 0
            elif isinstance(v, np.ndarray):
                fill = None if v.dtype == object else 0
                if index is None:
                    v[...] = fill
                else:
                    v[index] = fill
        return self

    # ---- split (batch.py:1209-1225) -----------------------------------------------------------
    def split(self, size: int, shuffle: bool = True, merge_last: bool = False) -> Iterator[Batch]:
        length = len(self)
        if size == -1:
            size = length
        assert size >= 1
        indices = np.random.permutation(length) if shuffle else np.arange(length)
        for lo, hi in split_bounds(length, size, merge_last):
            yield self[indices[lo:hi]]

    # ---- null checks (collector.py:512, trainer.py:928) ---------------------------------------
    def isnull(self) -> Batch:
        out = Batch()
        for k, v in self.items():
            if isinstance(v, Batch):
                out.__dict__[k] = v.isnull()
            elif isinstance(v, torch.Tensor):
                out.__dict__[k] = torch.isnan(v).cpu().numpy() if v.is_floating_point() else np.zeros(tuple(v.shape), bool)
            else:
                a = np.asarray(v)
                if a.dtype == object:
                    out.__dict__[k] = np.array([e is None for e in a.reshape(-1)]).reshape(a.shape)
                elif np.issubdtype(a.dtype, np.floating):
                    out.__dict__[k] = np.isnan(a)
                else:
                    out.__dict__[k] = np.zeros(a.shape, bool)
        return out

    def hasnull(self) -> bool:
        def any_true(b: Batch) -> bool:
            return any(any_true(v) if isinstance(v, Batch) else bool(np.any(v)) for v in b.values())

        return any_true(self.isnull())


def _zeros_like_value(v):
    """The "zero" of a leaf: 0 for numbers, None for objects, recursively for nested Batches."""
    if isinstance(v, Batch):
        return Batch({k: _zeros_like_value(x) for k, x in v.items()})
    if isinstance(v, torch.Tensor):
        return torch.zeros_like(v)
    a = np.asarray(v)
    if a.dtype == object:
        out = np.empty(a.shape, dtype=object)
        out[...] = None
        return out
    return np.zeros_like(a)


def _stack_np(arrs: list[np.ndarray], axis: int) -> np.ndarray:
    try:
        return np.stack(arrs, axis)
    except ValueError:
        out = np.empty(len(arrs), dtype=object)
        for i, a in enumerate(arrs):
            out[i] = a
        return out


def split_bounds(length: int, size: int, merge_last: bool = True) -> list[tuple[int, int]]:
    """Slice bounds of `Batch.split` (batch.py:1215-1225): with merge_last the final short remainder is
    folded into the previous minibatch (length 150, size 64 -> 64 + 86)."""
    if size == -1:
        size = length
    if size < 1 or length <= 0:
        return []
    merge_last = merge_last and length % size > 0
    out = []
    for idx in range(0, length, size):
        if merge_last and idx + size + size >= length:
            out.append((idx, length))
            break
        out.append((idx, min(idx + size, length)))
    return out


def to_numpy(x: Any) -> Any:
    """tianshou.data.utils.converter.to_numpy (converter.py:17-42) for the types used on the path."""
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    if isinstance(x, Batch):
        return x.to_numpy()
    if isinstance(x, dict):
        return {k: to_numpy(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return np.asanyarray([to_numpy(e) for e in x])
    return np.asanyarray(x) if x is not None else np.array(None, dtype=object)


def to_torch(x: Any, dtype: torch.dtype | None = None, device: str | torch.device = "cpu") -> Any:
    """converter.py:45-77."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype or x.dtype)
    if isinstance(x, Batch):
        return x.to_torch(dtype, device)
    if isinstance(x, dict):
        return {k: to_torch(v, dtype, device) for k, v in x.items()}
    a = np.asanyarray(x)
    if a.dtype == object:
        raise TypeError(f"object {x} cannot be converted to torch.")
    t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return t.to(dtype) if dtype is not None else t


def to_torch_as(x: Any, y: torch.Tensor) -> Any:
    return to_torch(x, dtype=y.dtype, device=y.device)
