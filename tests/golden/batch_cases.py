"""Scripted `Batch` operations of the hot path, written once and run twice: by
tests/golden/make_api_fixtures.py against the REFERENCE's `tianshou.data.Batch` (outputs -> batch_ops.npz) and by
tests/test_host_batch.py against the product's `tianshou_marl_amd.data.Batch`.  Every case returns a Batch (or a list of
Batches); `flatten` turns it into {dotted key path: ndarray} for comparison of keys, shapes, dtypes and values.
Operations: construction from per-env dicts (venvs.py:311-322 stacking), `__getitem__` (batch.py:723-747),
`__setitem__` (:778-802), `cat` / `stack` (:1045-1133), `split` incl. merge_last (:1209-1225), `to_torch` / `to_numpy`,
`empty_`, `update`, `len` / `shape`, the replay buffer's row format (obs / act / rew / flags / obs_next / info / policy)."""
import numpy as np


def flatten(b, prefix=""):
    out = {}
    if isinstance(b, (list, tuple)):
        for i, e in enumerate(b):
            out.update(flatten(e, f"{prefix}[{i}]."))
        return out
    for k in sorted(b.get_keys() if hasattr(b, "get_keys") else b.keys()):
        v = b[k]
        if hasattr(v, "get_keys"):
            if len(list(v.get_keys())) == 0:
                out[prefix + k + ".<empty>"] = np.zeros(0)
            out.update(flatten(v, prefix + k + "."))
        else:
            import torch

            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if a.dtype == object:
                a = np.array([str(x) for x in a.reshape(-1)]).reshape(a.shape)
            out[prefix + k] = a
    return out


def _rows(Batch, n=6, d=3):
    rng = np.random.default_rng(0)
    return Batch(obs=Batch(agent_id=np.array([f"agent_{i % 3}" for i in range(n)], dtype=object),
                           obs=rng.standard_normal((n, d)).astype(np.float32), mask=rng.random((n, 4)) < 0.7),
                 act=rng.integers(0, 4, n), rew=rng.standard_normal((n, 3)), terminated=rng.random(n) < 0.2,
                 truncated=np.zeros(n, bool), info=Batch(env_id=np.arange(n)), policy=Batch())


def cases(Batch):
    import torch

    out = {}
    # construction from what a vector env returns: an object array of per-env dicts (AEC and parallel layouts)
    aec = np.array([{"agent_id": f"agent_{i % 2}", "obs": np.full(2, i, np.float32), "mask": [True, i % 2 == 0]} for i in range(4)],
                   dtype=object)
    out["from_aec_dicts"] = Batch(obs=aec, info=np.array([{"env_id": i} for i in range(4)], dtype=object))
    par = np.array([{"observations": {"a0": np.full(3, i, np.float32), "a1": np.full(3, -i, np.float32)},
                     "agent_ids": ["a0", "a1"]} for i in range(3)], dtype=object)
    out["from_parallel_dicts"] = Batch(obs=par)
    b = _rows(Batch)
    out["getitem_int"] = b[2]
    out["getitem_slice"] = b[1:5:2]
    out["getitem_fancy"] = b[np.array([5, 0, 3])]
    out["getitem_bool"] = b[np.array([True, False, True, False, False, True])]
    c = _rows(Batch)
    c[np.array([0, 2])] = b[np.array([4, 5])]
    out["setitem_rows"] = c
    c2 = _rows(Batch)
    c2.act[1:3] = 9
    c2.obs.obs[0] = 7.0
    out["setitem_leaf"] = c2
    out["cat"] = Batch.cat([b[:2], b[4:], b[2:3]])
    out["stack"] = Batch.stack([b[0], b[3], b[5]])
    out["stack_axis1"] = Batch.stack([Batch(x=np.arange(6).reshape(2, 3)), Batch(x=-np.arange(6).reshape(2, 3))], axis=1)
    out["split_merge_last"] = list(b.split(4, shuffle=False, merge_last=True))
    out["split_plain"] = list(b.split(4, shuffle=False, merge_last=False))
    np.random.seed(3)
    out["split_shuffled"] = list(Batch(x=np.arange(10)).split(3, shuffle=True, merge_last=True))
    t = Batch(a=np.arange(4, dtype=np.float64), b=Batch(c=np.ones((4, 2), np.float32)))
    t.to_torch_(dtype=torch.float32)
    out["to_torch"] = t
    t2 = Batch(a=torch.arange(3), b=Batch(c=torch.ones(3, 2)))
    t2.to_numpy_()
    out["to_numpy"] = t2
    e = _rows(Batch)
    e.empty_(np.array([1, 4]))
    out["empty_rows"] = e
    u = Batch(a=np.arange(3))
    u.update(b=np.ones(3), c=Batch(d=np.zeros(3)))
    out["update"] = u
    out["meta"] = Batch(len_b=np.array(len(b)), shape_b=np.array(b.shape), len_nested=np.array(len(b.obs)),
                        empty_policy=np.array(len(b.policy.get_keys()) == 0), keys=np.array(sorted(b.get_keys()), dtype=object))
    return out
