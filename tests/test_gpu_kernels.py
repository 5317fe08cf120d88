"""GPU parity tests (`-m gpu`): the HIP path, called through the C-ABI (tianshou_marl_amd.ops ->
_abi -> libtsmarl_hip.so), against the CPU oracle and the committed golden fixtures.

Bars: bit-exact for integer / index work; for floating point the tolerance is written at each
assert (north_star: 1e-5 relative for GAE/returns).
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tianshou_marl_amd import ops

DEV = "cuda"


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=True)


def t(x, dtype=None):
    x = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(DEV)


# ------------------------------------------------------------------------------------------------
# GAE
# ------------------------------------------------------------------------------------------------
def _rand_gae_inputs(rng, T, L, p_term=0.02, p_trunc=0.02):
    v_s = rng.standard_normal((T, L)).astype(np.float32)
    v_n = rng.standard_normal((T, L)).astype(np.float32)
    rew = rng.standard_normal((T, L)).astype(np.float32)
    term = rng.random((T, L)) < p_term
    trunc = rng.random((T, L)) < p_trunc
    return v_s, v_n, rew, term, trunc


@pytest.mark.parametrize("T,L", [(1, 1), (1, 64), (7, 3), (8, 64), (9, 65), (25, 12), (25, 3072), (33, 200),
                                 (64, 1000), (129, 130), (257, 64), (1000, 7)])
def test_gae_matches_oracle(oracle, T, L):
    rng = np.random.default_rng(T * 1000 + L)
    v_s, v_n, rew, term, trunc = _rand_gae_inputs(rng, T, L)
    ret_o, adv_o = oracle.gae_lanes(v_s, v_n, rew, term, trunc, 0.99, 0.95)
    ret, adv = ops.gae_lanes(t(v_s), t(v_n), t(rew), t(term), t(trunc), 0.99, 0.95)
    # f64 accumulate on both sides, one f32 rounding on the device: 1e-6 relative (bar: 1e-5)
    np.testing.assert_allclose(adv.cpu().numpy(), adv_o, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ret.cpu().numpy(), ret_o, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("T,L,per_env", [(12800, 1, False), (1024, 1, False), (1500, 3, True), (5000, 64, False),
                                         (4097, 2, True), (20000, 5, False)])
def test_gae_few_long_lanes_take_the_time_parallel_kernel(oracle, T, L, per_env):
    """T >= 1024 with at most 64 lanes (the MARL trainers' per-agent batch is ONE lane of n_env * T rows): gae_long_kernel,
    one workgroup per lane, chunks combined by a parallel suffix scan of their affine maps.  Same bar as the lane kernel
    (1e-6 vs the f64 oracle); episode ends every ~25 steps, rare ones, and none at all (discount chains of 1000+ steps)."""
    for p_end in (0.04, 0.001, 0.0):
        rng = np.random.default_rng(T + L)
        v_s, v_n, rew, term, trunc = _rand_gae_inputs(rng, T, L, p_end, p_end)
        kw = {}
        if per_env:  # one flag per env shared by its lanes (here: every lane its own env group of size L)
            term, trunc = term[:, :1].copy(), trunc[:, :1].copy()
            term_l, trunc_l = np.repeat(term, L, axis=1), np.repeat(trunc, L, axis=1)
            kw = dict(lanes_per_env=L)
        else:
            term_l, trunc_l = term, trunc
        ret_o, adv_o = oracle.gae_lanes(v_s, v_n, rew, term_l, trunc_l, 0.99, 0.95, v_scale=1.7)
        ret, adv = ops.gae_lanes(t(v_s), t(v_n), t(rew), t(term), t(trunc), 0.99, 0.95, v_scale=1.7, **kw)
        np.testing.assert_allclose(adv.cpu().numpy(), adv_o, rtol=1e-6, atol=2e-6)
        np.testing.assert_allclose(ret.cpu().numpy(), ret_o, rtol=1e-6, atol=2e-6)


@pytest.mark.parametrize("T,L,per_env", [(4097, 1, False), (12800, 1, False), (12800, 2, True), (20000, 5, False), (33000, 3, False)])
def test_gae_long_lanes_super_chunks_side_by_side_equal_the_sequential_scan(oracle, T, L, per_env):
    """Above 4096 steps a long lane is several super-chunks.  With the scan workspace registered (ops.ensure_scan_workspace) they run
    on different workgroups and hand their affine maps over through device memory; without it one workgroup walks them in turn.
    Same recurrence, same order: identical bits -- also on repeated launches (the workspace's generation word advances per launch)
    and as a hipGraph replay -- and both within 1e-6 of the f64 oracle."""
    from tianshou_marl_amd import _abi

    rng = np.random.default_rng(T + L)
    v_s, v_n, rew, term, trunc = _rand_gae_inputs(rng, T, L, 0.01, 0.01)
    kw = {}
    term_l, trunc_l = term, trunc
    if per_env:
        term, trunc = term[:, :1].copy(), trunc[:, :1].copy()
        term_l, trunc_l = np.repeat(term, L, axis=1), np.repeat(trunc, L, axis=1)
        kw = dict(lanes_per_env=L)
    args = (t(v_s), t(v_n), t(rew), t(term), t(trunc), 0.99, 0.95)
    assert ops.ensure_scan_workspace(DEV)
    n = int(_abi.call("tsm_gae_scan_workspace_bytes"))
    try:
        _abi.call("tsm_gae_set_scan_workspace", None, 0)  # withdrawn: the sequential form
        ret_s, adv_s = ops.gae_lanes(*args, v_scale=1.7, **kw)
    finally:
        _abi.call("tsm_gae_set_scan_workspace", ops._scan_ws[torch.cuda.current_device()][0].data_ptr(), n)
    for _ in range(3):
        ret_p, adv_p = ops.gae_lanes(*args, v_scale=1.7, **kw)
        assert torch.equal(ret_p, ret_s) and torch.equal(adv_p, adv_s)
    out = (torch.empty_like(ret_s), torch.empty_like(adv_s))
    ops.gae_lanes(*args, v_scale=1.7, out=out, **kw)
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        ops.gae_lanes(*args, v_scale=1.7, out=out, **kw)
    for _ in range(3):
        out[0].zero_(); out[1].zero_()
        g.replay()
        assert torch.equal(out[0], ret_s) and torch.equal(out[1], adv_s)
    # two launches in flight on two streams take different slots of the workspace (they used to share one)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1, o2 = (torch.empty_like(ret_s), torch.empty_like(adv_s)), (torch.empty_like(ret_s), torch.empty_like(adv_s))
    torch.cuda.synchronize()
    for _ in range(4):
        with torch.cuda.stream(s1):
            ops.gae_lanes(*args, v_scale=1.7, out=o1, **kw)
        with torch.cuda.stream(s2):
            ops.gae_lanes(*args, v_scale=1.7, out=o2, **kw)
    torch.cuda.synchronize()
    assert torch.equal(o1[0], ret_s) and torch.equal(o2[0], ret_s) and torch.equal(o1[1], adv_s) and torch.equal(o2[1], adv_s)
    assert not ops.gae_scan_failed()
    ret_o, adv_o = oracle.gae_lanes(v_s, v_n, rew, term_l, trunc_l, 0.99, 0.95, v_scale=1.7)
    np.testing.assert_allclose(adv_p.cpu().numpy(), adv_o, rtol=1e-6, atol=2e-6)
    np.testing.assert_allclose(ret_p.cpu().numpy(), ret_o, rtol=1e-6, atol=2e-6)


def test_gae_env_level_flags_and_return_scaling(oracle):
    rng = np.random.default_rng(3)
    T, n_env, N = 25, 37, 3
    v_s, v_n, rew, _, _ = _rand_gae_inputs(rng, T, n_env * N)
    term_e = rng.random((T, n_env)) < 0.05
    trunc_e = rng.random((T, n_env)) < 0.05
    term_l, trunc_l = np.repeat(term_e, N, axis=1), np.repeat(trunc_e, N, axis=1)
    scale = 2.37
    ret_o, adv_o = oracle.gae_lanes(v_s, v_n, rew, term_l, trunc_l, 0.97, 0.9, v_scale=scale)
    ret, adv = ops.gae_lanes(t(v_s), t(v_n), t(rew), t(term_e), t(trunc_e), 0.97, 0.9, v_scale=scale,
                             lanes_per_env=N)
    np.testing.assert_allclose(adv.cpu().numpy(), adv_o, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ret.cpu().numpy(), ret_o, rtol=1e-6, atol=1e-6)


def test_gae_reference_fixture(golden_dir):
    """Outputs of the reference's compute_episodic_return (tests/golden/gae.npz) in lane layout."""
    g = _load(golden_dir, "gae.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        n_env, T = int(g[p + "n_env"]), int(g[p + "T"])
        L = lambda x: np.ascontiguousarray(np.asarray(x).reshape(n_env, T).T)  # noqa: E731
        ret, adv = ops.gae_lanes(t(L(g[p + "v_s"])), t(L(g[p + "v_s_next"])), t(L(g[p + "rew"]), torch.float32),
                                 t(L(g[p + "terminated"])), t(L(g[p + "truncated"])),
                                 float(g[p + "gamma"]), float(g[p + "lam"]))
        # rew is f64 in the reference and f32 on the device: 1e-5 relative (the north_star bar)
        np.testing.assert_allclose(adv.cpu().numpy().T.reshape(-1), g[p + "adv"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ret.cpu().numpy().T.reshape(-1), g[p + "returns"], rtol=1e-5, atol=1e-5)


def test_gae_ragged_rotated_subbuffers(oracle):
    """env_start / env_len: circular sub-buffers with different fill levels (empty ones included)."""
    rng = np.random.default_rng(11)
    T, n_env, N = 20, 9, 2
    L = n_env * N
    v_s, v_n, rew, term, trunc = _rand_gae_inputs(rng, T, L, 0.05, 0.05)
    start = rng.integers(0, T, n_env).astype(np.int32)
    length = rng.integers(0, T + 1, n_env).astype(np.int32)
    length[0], length[1] = 0, T
    ret, adv = ops.gae_lanes(t(v_s), t(v_n), t(rew), t(term), t(trunc), 0.99, 0.95, lanes_per_env=N,
                             env_start=t(start), env_len=t(length),
                             out=(torch.full((T, L), -7.0, device=DEV), torch.full((T, L), -7.0, device=DEV)))
    ret, adv = ret.cpu().numpy(), adv.cpu().numpy()
    for e in range(n_env):
        slots = (start[e] + np.arange(length[e])) % T
        untouched = np.setdiff1d(np.arange(T), slots)
        for a in range(N):
            l = e * N + a
            assert np.all(adv[untouched, l] == -7.0) and np.all(ret[untouched, l] == -7.0)
            if length[e] == 0:
                continue
            r_o, a_o = oracle.gae_lanes(v_s[slots, l][:, None], v_n[slots, l][:, None], rew[slots, l][:, None],
                                        term[slots, l][:, None], trunc[slots, l][:, None], 0.99, 0.95)
            np.testing.assert_allclose(adv[slots, l], a_o[:, 0], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(ret[slots, l], r_o[:, 0], rtol=1e-6, atol=1e-6)


def test_gae_empty_and_errors():
    e = torch.empty(0, 8, device=DEV)
    ret, adv = ops.gae_lanes(e, e, e, e.to(torch.uint8), e.to(torch.uint8))
    assert ret.shape == (0, 8)
    x = torch.zeros(4, 7, device=DEV)
    with pytest.raises(ValueError):
        ops.gae_lanes(x, x, x, x.to(torch.uint8), x.to(torch.uint8), lanes_per_env=2)
    with pytest.raises(ValueError):
        ops.gae_lanes(x, x, x, x.to(torch.uint8), x.to(torch.uint8), v_scale=0.0)


def test_gae_full_size_properties(oracle):
    """BASELINE config C3 (n_env=4096, n_agent=8, T=25) against the oracle, and size-independent
    properties at a long horizon: linearity in (rew, v) and returns - adv == v_s."""
    rng = np.random.default_rng(5)
    T, L = 25, 4096 * 8
    v_s, v_n, rew, term, trunc = _rand_gae_inputs(rng, T, L, 0.01, 0.0)
    trunc[-1] = True
    ret_o, adv_o = oracle.gae_lanes(v_s, v_n, rew, term, trunc, 0.99, 0.95, threads=8)
    ret, adv = ops.gae_lanes(t(v_s), t(v_n), t(rew), t(term), t(trunc))
    np.testing.assert_allclose(adv.cpu().numpy(), adv_o, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ret.cpu().numpy(), ret_o, rtol=1e-6, atol=1e-6)
    T, L = 2048, 4096
    g = torch.Generator(device=DEV).manual_seed(0)
    mk = lambda: torch.randn(T, L, device=DEV, generator=g)  # noqa: E731
    v1, n1, r1, v2, n2, r2 = mk(), mk(), mk(), mk(), mk(), mk()
    te = (torch.rand(T, L, device=DEV, generator=g) < 0.01).to(torch.uint8)
    tr = (torch.rand(T, L, device=DEV, generator=g) < 0.01).to(torch.uint8)
    ret1, adv1 = ops.gae_lanes(v1, n1, r1, te, tr)
    ret2, adv2 = ops.gae_lanes(v2, n2, r2, te, tr)
    ret3, adv3 = ops.gae_lanes(v1 + 2 * v2, n1 + 2 * n2, r1 + 2 * r2, te, tr)
    assert torch.allclose(adv3, adv1 + 2 * adv2, rtol=1e-4, atol=1e-4)
    assert torch.allclose(ret1 - adv1, v1, rtol=1e-5, atol=1e-5)
    # a spot lane against the oracle at the long horizon
    ls = slice(100, 164)
    c = lambda x: x[:, ls].cpu().numpy()  # noqa: E731
    r_o, a_o = oracle.gae_lanes(c(v1), c(n1), c(r1), c(te), c(tr))
    np.testing.assert_allclose(c(adv1), a_o, rtol=1e-6, atol=1e-6)


def test_mc_return_to_go(oracle):
    rng = np.random.default_rng(2)
    rew = rng.standard_normal((17, 5)).astype(np.float32)
    out = ops.mc_return_to_go_lanes(t(rew), 0.97).cpu().numpy()
    for l in range(5):
        np.testing.assert_allclose(out[:, l], oracle.episode_mc_return_to_go(rew[:, l], 0.97), rtol=1e-6)


# ------------------------------------------------------------------------------------------------
# VectorReplayBuffer index algebra: bit-exact
# ------------------------------------------------------------------------------------------------
def test_vrb_trace_bit_exact(golden_dir):
    g = _load(golden_dir, "vrb_trace.npz")
    for ci in range(int(g["n_cfg"])):
        p = f"c{ci}_"
        total, num, rew_dim = (int(x) for x in g[p + "cfg"])
        buf = ops.VrbState(total, num, rew_dim)
        S = buf.sub_size
        payload = torch.zeros(S, num, 3, dtype=torch.float32, device=DEV)
        for ai in range(int(g[p + "n_add"])):
            if p + "reset_at" in g and ai == int(g[p + "reset_at"]):
                buf.reset(keep_statistics=True)
            q = f"{p}a{ai}_"
            done = g[q + "term"] | g[q + "trunc"]
            rew = g[q + "rew"].astype(np.float32)
            k = len(done)
            src = torch.full((k, 3), float(ai), device=DEV) + t(g[q + "ids"], torch.float32)[:, None]
            ptr, ep_rew, ep_len, ep_idx = buf.add(t(rew), t(done), t(g[q + "ids"]), fields=[(src, payload)])
            ptr_h = ptr.cpu().numpy()
            assert np.array_equal(ptr_h, g[q + "ptr"]), (ci, ai)
            assert np.array_equal(ep_len.cpu().numpy(), g[q + "ep_len"]), (ci, ai)
            assert np.array_equal(ep_idx.cpu().numpy(), g[q + "ep_idx"]), (ci, ai)
            assert np.array_equal(ep_rew.cpu().numpy().reshape(g[q + "ep_rew"].shape), g[q + "ep_rew"]), (ci, ai)
            assert len(buf) == int(g[q + "len"])
            assert np.array_equal(buf.unfinished_index().cpu().numpy(), g[q + "unfinished"]), (ci, ai)
            assert np.array_equal(buf.sample_indices_all().cpu().numpy(), g[q + "sample0"]), (ci, ai)
            # payload landed at (slot, env) of the time-major store == flat index env*S+slot
            got = buf.gather(payload, ptr)
            assert torch.equal(got, src)
            if q + "prev" in g:
                allidx = t(np.arange(-2, buf.maxsize + 2))
                assert np.array_equal(buf.prev(allidx).cpu().numpy(), g[q + "prev"]), (ci, ai)
                assert np.array_equal(buf.next(allidx).cpu().numpy(), g[q + "next"]), (ci, ai)
                done_flat = buf.done_store.t().reshape(-1).cpu().numpy().astype(bool)
                assert np.array_equal(done_flat, g[q + "done"].astype(bool))
                assert np.array_equal(buf.last_index.cpu().numpy(), g[q + "last_index"])
        buf.check()


def test_vrb_reference_known_answers():
    """test/base/test_buffer.py:740-964 replayed on the device buffer."""
    buf = ops.VrbState(20, 4)
    f = lambda x: t(np.asarray(x, np.float32))  # noqa: E731
    b = lambda x: t(np.asarray(x, bool))  # noqa: E731
    i = lambda x: t(np.asarray(x, np.int64))  # noqa: E731
    ptr, ep_rew, ep_len, ep_idx = buf.add(f([1, 2, 3]), b([0, 0, 1]), i([0, 1, 2]))
    assert ep_len.tolist() == [0, 0, 1] and ep_rew[:, 0].tolist() == [0, 0, 3]
    assert ptr.tolist() == [0, 5, 10] and ep_idx.tolist() == [0, 5, 10]
    idx = buf.sample_indices_all()
    assert idx.tolist() == [0, 5, 10]
    assert buf.prev(idx).tolist() == [0, 5, 10] and buf.next(idx).tolist() == [0, 5, 10]
    assert buf.unfinished_index().tolist() == [0, 5]
    buf.add(f([4]), b([1]), i([3]))
    assert buf.unfinished_index().tolist() == [0, 5]
    z = np.zeros(4)
    buf.add(f(z), b(z), i([0, 1, 2, 3]))
    buf.add(f(z), b(1 - z), i([0, 1, 2, 3]))
    assert len(buf) == 12
    buf.add(f(z), b(z), i([0, 1, 2, 3]))
    buf.add(f(z), b([0, 1, 0, 1]), i([0, 1, 2, 3]))
    assert len(buf) == 20
    idx = buf.sample_indices_all()
    assert idx.tolist() == list(range(20))
    assert buf.prev(idx).tolist() == [0, 0, 1, 3, 3, 5, 5, 6, 8, 8, 10, 11, 11, 13, 13, 15, 16, 16, 18, 18]
    assert buf.next(idx).tolist() == [1, 2, 2, 4, 4, 6, 7, 7, 9, 9, 10, 12, 12, 14, 14, 15, 17, 17, 19, 19]
    assert buf.unfinished_index().tolist() == [4, 14]
    ptr, ep_rew, ep_len, ep_idx = buf.add(f([1]), b([1]), i([2]))
    assert (ep_len.tolist(), ep_rew[:, 0].tolist(), ptr.tolist(), ep_idx.tolist()) == ([3], [1], [10], [13])
    assert buf.unfinished_index().tolist() == [4]
    idx = torch.sort(buf.sample_indices_all()).values
    assert buf.prev(idx).tolist() == [0, 0, 1, 3, 3, 5, 5, 6, 8, 8, 14, 11, 11, 13, 13, 15, 16, 16, 18, 18]
    assert buf.next(idx).tolist() == [1, 2, 2, 4, 4, 6, 7, 7, 9, 9, 10, 12, 12, 14, 10, 15, 17, 17, 19, 19]
    assert buf.prev(i([-1])).tolist() == buf.prev(i([19])).tolist()


def test_vrb_large_scatter_matches_oracle(oracle):
    """C2-sized vector steps (1024 envs, N=3, obs 18): every field lands where the reference index says."""
    n_env, N, D, T = 1024, 3, 18, 25
    buf = ops.VrbState(n_env * T, n_env, N)
    ob = oracle.VectorReplayBufferIndex(n_env * T, n_env, N)
    obs_store = torch.zeros(T, n_env, N, D, device=DEV)
    act_store = torch.zeros(T, n_env, N, dtype=torch.int32, device=DEV)
    flag_store = torch.zeros(T, n_env, N, dtype=torch.uint8, device=DEV)
    rng = np.random.default_rng(0)
    for step in range(T + 3):  # wraps around
        obs = torch.randn(n_env, N, D, device=DEV)
        act = torch.randint(0, 5, (n_env, N), dtype=torch.int32, device=DEV)
        flg = torch.randint(0, 2, (n_env, N), dtype=torch.uint8, device=DEV)
        rew = rng.integers(-2, 3, (n_env, N)).astype(np.float32)
        done = rng.random(n_env) < 0.05
        ptr, ep_rew, ep_len, ep_idx = buf.add(t(rew), t(done), None,
                                              fields=[(obs, obs_store), (act, act_store), (flg, flag_store)])
        p_o, r_o, l_o, i_o = ob.add(rew.astype(np.float64), done)
        assert np.array_equal(ptr.cpu().numpy(), p_o)
        assert np.array_equal(ep_rew.cpu().numpy(), r_o)
        assert np.array_equal(ep_len.cpu().numpy(), l_o) and np.array_equal(ep_idx.cpu().numpy(), i_o)
        assert torch.equal(buf.gather(obs_store, ptr), obs)
        assert torch.equal(buf.gather(act_store, ptr), act)
        assert torch.equal(buf.gather(flag_store, ptr), flg)
    assert np.array_equal(buf.sample_indices_all().cpu().numpy(), ob.sample_indices_all())
    assert np.array_equal(buf.unfinished_index().cpu().numpy(), ob.unfinished_index())


# ------------------------------------------------------------------------------------------------
# agent dispatch: bit-exact with numpy nonzero
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,n_agent", [(0, 3), (1, 1), (37, 3), (1024, 8), (5000, 5), (100000, 3)])
def test_agent_index_matches_numpy(oracle, B, n_agent):
    rng = np.random.default_rng(B + n_agent)
    ids = rng.integers(0, n_agent, B).astype(np.int32)
    index, offsets = ops.agent_index(t(ids), n_agent)
    index, offsets = index.cpu().numpy(), offsets.cpu().numpy()
    for a in range(n_agent):
        assert np.array_equal(index[offsets[a]:offsets[a + 1]], oracle.agent_index(ids, a))
    assert offsets[-1] == B


def test_marl_dispatch_fixture(golden_dir):
    g = _load(golden_dir, "marl_dispatch.npz")
    rows, obs, Ws = g["agent_rows"], g["obs"], g["Ws"]
    index, offsets = ops.agent_index(t(rows, torch.int32), 3)
    off = offsets.cpu().numpy()
    holder = torch.zeros(len(rows), dtype=torch.int32, device=DEV)
    obs_d = t(obs)
    for a in range(3):
        idx = index[off[a]:off[a + 1]]
        sub = ops.gather_rows(obs_d, idx)                       # batch[agent_index]
        act = torch.argmax(sub @ t(Ws[a]), -1).to(torch.int32)  # the mock policy of the fixture
        ops.scatter_rows(act, idx, holder)                      # holder.act[agent_index] = act
    assert np.array_equal(holder.cpu().numpy(), g["act_independent"])


# ------------------------------------------------------------------------------------------------
# categorical head
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,A", [(1, 2), (100, 5), (3072, 5), (1000, 17)])
def test_categorical_logp_entropy(oracle, B, A):
    rng = np.random.default_rng(B * A)
    logits = (rng.standard_normal((B, A)) * 3).astype(np.float32)
    act = rng.integers(0, A, B)
    lp_o, en_o = oracle.categorical_logp_entropy(logits, act)
    lp, en = ops.categorical_logp_entropy(t(logits), t(act, torch.int32))
    np.testing.assert_allclose(lp.cpu().numpy(), lp_o, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(en.cpu().numpy(), en_o, rtol=1e-5, atol=1e-6)


def test_categorical_sample_distribution_and_determinism(oracle):
    A, B = 5, 200000
    logits = np.tile(np.array([[0.1, -1.0, 2.0, 0.5, -0.3]], np.float32), (B, 1))
    act, logp = ops.categorical_sample(t(logits), seed=1234, offset=0)
    act2, _ = ops.categorical_sample(t(logits), seed=1234, offset=0)
    act3, _ = ops.categorical_sample(t(logits), seed=1235, offset=0)
    assert torch.equal(act, act2) and not torch.equal(act, act3)
    # counter-based: rows [off, off+n) of a long call equal a short call with that offset
    act4, _ = ops.categorical_sample(t(logits[:1000]), seed=1234, offset=5000)
    assert torch.equal(act4, act[5000:6000])
    a = act.cpu().numpy()
    p = np.exp(logits[0] - logits[0].max())
    p /= p.sum()
    counts = np.bincount(a, minlength=A)
    chi2 = ((counts - B * p) ** 2 / (B * p)).sum()
    assert chi2 < 30.0, chi2  # 4 dof, p ~ 5e-6
    lp_o, _ = oracle.categorical_logp_entropy(logits, a)
    np.testing.assert_allclose(logp.cpu().numpy(), lp_o, rtol=1e-5, atol=1e-6)
    mode, _ = ops.categorical_sample(t(logits), seed=0, deterministic=True)
    assert torch.all(mode == 2)


# ------------------------------------------------------------------------------------------------
# PPO loss
# ------------------------------------------------------------------------------------------------
def _rand_ppo(rng, M, A):
    logits = rng.standard_normal((M, A)).astype(np.float32)
    act = rng.integers(0, A, M)
    logp_old = (rng.standard_normal(M) * 0.3 - 1.5).astype(np.float32)
    adv = (rng.standard_normal(M) * 2 + 0.3).astype(np.float32)
    returns = rng.standard_normal(M).astype(np.float32)
    value = rng.standard_normal(M).astype(np.float32)
    v_old = (value + rng.standard_normal(M) * 0.3).astype(np.float32)
    return logits, act, logp_old, adv, returns, value, v_old


@pytest.mark.parametrize("M,A", [(2, 5), (64, 5), (257, 5), (4096, 5), (1000, 3), (300, 9)])
@pytest.mark.parametrize("variant", ["default", "dual_vclip", "nonorm"])
def test_ppo_loss_matches_oracle(oracle, M, A, variant):
    rng = np.random.default_rng(M + A)
    logits, act, logp_old, adv, returns, value, v_old = _rand_ppo(rng, M, A)
    kw = dict(default=dict(), dual_vclip=dict(dual_clip=2.0, value_clip=True, eps_clip=0.1),
              nonorm=dict(adv_norm=False, vf_coef=0.25, ent_coef=0.02))[variant]
    o = oracle.ppo_loss(logits, act, logp_old, adv, returns, value, v_old, **kw)
    cfg = ops.make_ppo_cfg(**kw)
    stats = ops.ppo_adv_stats(t(adv), t(np.array([0, M], np.int64)))
    dl, dv, sc = ops.ppo_loss_fwd_bwd(t(logits), t(value), t(act, torch.int32), t(logp_old), t(adv), t(returns),
                                      cfg, adv_stats=stats[0], v_s_old=t(v_old))
    if kw.get("adv_norm", True):
        np.testing.assert_allclose(stats.cpu().numpy()[0], [o["adv_mean"], o["adv_std"]], rtol=1e-6)
    sc = sc.cpu().numpy()
    # f32 per-sample math vs f64 oracle: 1e-5 relative on the scalars, 1e-4 on gradients
    np.testing.assert_allclose(sc, [o["loss"], o["clip_loss"], o["vf_loss"], o["ent_loss"]], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dl.cpu().numpy(), o["dlogits"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(dv.cpu().numpy(), o["dvalue"], rtol=1e-4, atol=1e-7)


def test_ppo_loss_minibatch_permutation(oracle):
    """Minibatches addressed through a permutation (Batch.split, batch.py:1219) + per-minibatch stats."""
    rng = np.random.default_rng(0)
    n, A = 1000, 5
    logits, act, logp_old, adv, returns, value, v_old = _rand_ppo(rng, n, A)
    perm = rng.permutation(n)
    bounds = oracle.split_bounds(n, 300, True)  # 300, 300, 400
    assert [e - s for s, e in bounds] == [300, 300, 400]
    mb_start = np.array([s for s, _ in bounds] + [n], np.int64)
    stats = ops.ppo_adv_stats(t(adv), t(mb_start), perm=t(perm))
    cfg = ops.make_ppo_cfg()
    for k, (s, e) in enumerate(bounds):
        rows = perm[s:e]
        o = oracle.ppo_loss(logits[rows], act[rows], logp_old[rows], adv[rows], returns[rows], value[rows])
        dl, dv, sc = ops.ppo_loss_fwd_bwd(t(logits[rows]), t(value[rows]), t(act, torch.int32), t(logp_old), t(adv),
                                          t(returns), cfg, adv_stats=stats[k], perm=t(perm[s:e]))
        np.testing.assert_allclose(sc.cpu().numpy(), [o["loss"], o["clip_loss"], o["vf_loss"], o["ent_loss"]],
                                   rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dl.cpu().numpy(), o["dlogits"], rtol=1e-4, atol=1e-7)


def test_ppo_loss_reference_fixture(golden_dir):
    """Loss scalars produced by the reference PPO on the full batch (tests/golden/ppo_update.npz)."""
    g = _load(golden_dir, "ppo_update.npz")
    for name in g["variants"]:
        p = str(name) + "_"
        if int(g[p + "batch_size"]) != -1:
            continue
        eps_clip, dual_clip, value_clip, adv_norm, vf_coef, ent_coef = g[p + "ppo_cfg"][:6]
        cfg = ops.make_ppo_cfg(eps_clip, dual_clip or None, bool(value_clip), bool(adv_norm), vf_coef, ent_coef)
        n = len(g[p + "act"])
        stats = ops.ppo_adv_stats(t(g[p + "adv"]), t(np.array([0, n], np.int64)))
        _, _, sc = ops.ppo_loss_fwd_bwd(t(g[p + "logits"]), t(g[p + "v_s"]), t(g[p + "act"], torch.int32),
                                        t(g[p + "logp_old"]), t(g[p + "adv"]), t(g[p + "returns"]), cfg,
                                        adv_stats=stats[0], v_s_old=t(g[p + "v_s"]))
        ref = [g[p + "stat_loss"][0], g[p + "stat_actor_loss"][0], g[p + "stat_vf_loss"][0], g[p + "stat_ent_loss"][0]]
        np.testing.assert_allclose(sc.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
        lp, _ = ops.categorical_logp_entropy(t(g[p + "logits"]), t(g[p + "act"], torch.int32))
        np.testing.assert_allclose(lp.cpu().numpy(), g[p + "logp_old"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------
# optimizer step vs torch.optim.Adam + clip_grad_norm_ (the third-party arithmetic the reference calls)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("max_norm", [None, 0.5])
@pytest.mark.parametrize("n_slab", [1, 7])
def test_adam_step_matches_torch(max_norm, n_slab):
    torch.manual_seed(0)
    n = 11142
    p0 = torch.randn(n)
    slabs = torch.randn(3, n_slab, n) * 0.1
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=3e-4)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(3):
        ref.grad = slabs[step].sum(0)
        if max_norm:
            torch.nn.utils.clip_grad_norm_([ref], max_norm)
        opt.step()
        ops.adam_step(p, slabs[step].to(DEV), m, v, step + 1, lr=3e-4, max_grad_norm=max_norm)
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-7)


@pytest.mark.parametrize("max_norm", [None, 0.5])
def test_adam_step_over_slab_segments(max_norm):
    """`tsm_adam_step_segs`: actor and critic gradients arrive in slab arrays of their own (csrc/ppo_rows.hip), with
    different slab counts and pitches; one launch must equal what the unsegmented entry points do -- without clipping, two
    `adam_step` calls on the halves (bit for bit: the same slab order per parameter); with clipping, ONE norm over the whole
    vector as `clip_grad_norm_` over ActorCritic.parameters() computes it (vs torch)."""
    torch.manual_seed(1)
    n_a, n_c = 23429, 65921  # not multiples of the 64-parameter blocks
    n = n_a + n_c
    sl_a = (torch.randn(37, n_a) * 0.1).to(DEV)
    wide = (torch.randn(5, n_c + 77) * 0.1).to(DEV)   # a view into wider slabs: pitch > parameter count
    sl_c = wide[:, :n_c]
    p0 = torch.randn(n)
    segs = [(sl_a, 0, n_a), (wide, n_a, n_c)]
    g = ops.reduce_slabs_segs(segs, n, scale=0.5)
    assert torch.equal(g[:n_a], ops.reduce_slabs(sl_a, scale=0.5))
    assert torch.equal(g[n_a:], ops.reduce_slabs(sl_c.contiguous(), scale=0.5))
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step_dev = torch.tensor([0], dtype=torch.int64, device=DEV)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=3e-4)
    q, qm, qv = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(3):
        step_dev += 1
        ops.adam_step_segs(p, segs, m, v, 1, lr=3e-4, max_grad_norm=max_norm, step_dev=step_dev)
        ref.grad = torch.cat([sl_a.sum(0), sl_c.sum(0)]).cpu()
        if max_norm:
            torch.nn.utils.clip_grad_norm_([ref], max_norm)
        opt.step()
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-7)
        if not max_norm:
            ops.adam_step(q[:n_a], sl_a, qm[:n_a], qv[:n_a], step + 1, lr=3e-4)
            ops.adam_step(q[n_a:], sl_c.contiguous(), qm[n_a:], qv[n_a:], step + 1, lr=3e-4)
            assert torch.equal(p, q) and torch.equal(m, qm) and torch.equal(v, qv)
    with pytest.raises(ValueError):  # a gap between the segments
        ops.adam_step_segs(p, [(sl_a, 0, n_a), (wide, n_a + 1, n_c - 1)], m, v, 1)


# ------------------------------------------------------------------------------------------------
# CTDE global state
# ------------------------------------------------------------------------------------------------
def test_global_state_fixture(golden_dir):
    g = _load(golden_dir, "ctde.npz")
    oba = [t(x) for x in g["obs_by_agent"]]
    assert np.array_equal(ops.global_state(oba, "concatenate").cpu().numpy(), g["global_concatenate"])
    np.testing.assert_allclose(ops.global_state(oba, "mean").cpu().numpy(), g["global_mean"], rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError):
        ops.global_state(oba, "attention")


# ------------------------------------------------------------------------------------------------
# minibatch permutations (batch.py:1219 on the device)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 255, 1000, 25600, 76800, 819200])
def test_random_permutations_are_permutations(n):
    n_perm = 5 if n <= 76800 else 2
    out = ops.random_permutations(n, n_perm, seed=7, counter=11)
    assert out.shape == (n_perm, n) and out.dtype == torch.int64
    srt = torch.sort(out, dim=1).values
    assert torch.equal(srt, torch.arange(n, device=DEV).expand(n_perm, n))  # bijection of [0, n): sortedness property
    if n >= 64:
        assert not torch.equal(out[0], out[1])  # different draws
        assert not torch.equal(out[0], torch.arange(n, device=DEV))
    again = ops.random_permutations(n, n_perm, seed=7, counter=11)
    assert torch.equal(out, again)  # counter-based: reproducible
    if n >= 64:
        assert not torch.equal(out, ops.random_permutations(n, n_perm, seed=8, counter=11))


def test_random_permutations_affine_map_device_counter_and_uniformity():
    n, N, repeat = 400, 3, 2
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    out = ops.random_permutations(n, N * repeat, seed=3, counter_dev=ctr, scale=N, group_size=repeat, offset_mul=1)
    for p in range(N * repeat):  # lane ids of agent p // repeat: row * N + agent, each row exactly once
        a = p // repeat
        assert torch.equal(torch.sort(out[p]).values, torch.arange(n, device=DEV) * N + a)
    ops.call("tsm_u64_add", ops.ptr(ctr), N * repeat, ops.stream_ptr())
    nxt = ops.random_permutations(n, N * repeat, seed=3, counter_dev=ctr, scale=N, group_size=repeat, offset_mul=1)
    assert not torch.equal(out, nxt)  # the device counter advances the stream (graph replays draw fresh permutations)
    assert torch.equal(nxt, ops.random_permutations(n, N * repeat, seed=3, counter=N * repeat, scale=N, group_size=repeat,
                                                    offset_mul=1))
    # uniformity: where element 0 lands, and which element lands first, over 4000 draws of a 50-permutation
    m, draws = 50, 4000
    P = ops.random_permutations(m, draws, seed=5).cpu().numpy()
    for counts in (np.bincount(P[:, 0], minlength=m), np.bincount(np.argmax(P == 0, axis=1), minlength=m)):
        chi2 = ((counts - draws / m) ** 2 / (draws / m)).sum()
        assert chi2 < 100.0, chi2  # 49 dof: mean 49, 99.99th percentile ~ 94
    # pairs are decorrelated: P(pi(0) < pi(1)) ~ 1/2
    frac = (P[:, 0] < P[:, 1]).mean()
    assert abs(frac - 0.5) < 0.04
    with pytest.raises(ValueError):
        ops.random_permutations(-1, 1, seed=0)


def test_gather_fields_maps_and_converts_like_torch():
    """tsm_gather_fields (csrc/gather_fields.hip): env-major rows out of a time-major store with an agent's column offset, and
    converting copies, several fields per launch -- against the torch expressions they replace (exact: moves and integer / flag
    conversions only)."""
    T, E, N, D = 7, 5, 3, 6
    g = torch.Generator(device="cpu").manual_seed(0)
    obs = torch.randn(T, E, N, D, generator=g).to(DEV)
    act = torch.randint(0, 5, (T, E, N), generator=g, dtype=torch.int32).to(DEV)
    flag = (torch.rand(T, E, N, generator=g) < 0.3).to(torch.uint8).to(DEV)
    for a in range(N):
        o = torch.empty(E * T, D, device=DEV)
        ac = torch.empty(E * T, dtype=torch.int64, device=DEV)
        fb = torch.empty(E * T, dtype=torch.bool, device=DEV)
        ff = torch.empty(E * T, dtype=torch.float32, device=DEV)
        ops.gather_fields([(obs, o, T, E, N * D, a * D), (act, ac, T, E, N, a), (flag, fb, T, E, N, a), (flag, ff, T, E, N, a)])
        em = lambda x: x.transpose(0, 1)[:, :, a].reshape(E * T, *x.shape[3:])  # noqa: E731
        assert torch.equal(o, em(obs)) and torch.equal(ac, em(act).to(torch.int64))
        assert torch.equal(fb, em(flag).bool()) and torch.equal(ff, em(flag).float())
    # converting copies (what learn() does with a device batch): int64 -> int32, bool -> uint8, f32 -> f32; ten fields = two launches
    a64 = torch.randint(0, 5, (33,), generator=g).to(DEV)
    b = (torch.rand(33, generator=g) < 0.5).to(DEV)
    x = torch.randn(33, 4, generator=g).to(DEV)
    outs = [(a64, torch.empty(33, dtype=torch.int32, device=DEV)), (b, torch.empty(33, 1, dtype=torch.uint8, device=DEV)),
            (x, torch.empty(33, 4, device=DEV))] * 3 + [(a64, torch.empty(33, dtype=torch.int64, device=DEV))]
    ops.gather_fields(outs)
    for src, dst in outs:
        assert torch.equal(dst.reshape(src.shape).to(src.dtype), src)
    with pytest.raises(TypeError):
        ops.gather_fields([(x.double(), torch.empty(33, 4, device=DEV))])
    with pytest.raises(ValueError):
        ops.gather_fields([(obs, torch.empty(E * T, D, device=DEV), T, E, N * D, N * D)])  # the last agent's column + D leaves the store


@pytest.mark.parametrize("case", [
    (18, 5, 76800, 4096, {}, True, False),                                   # the job's gradient step: one tile per workgroup
    (18, 5, 76800, 5120, {}, True, False),                                   # two tiles on some workgroups (last-tile stores)
    (18, 5, 5000, 1000, {}, False, False),                                   # weights from the flat vector
    (48, 5, 9000, 4099, dict(dual_clip=2.0, value_clip=True), True, True),  # ragged tail, value clip
    (33, 9, 3000, 257, dict(adv_norm=False), True, False),
    (18, 5, 819200, 65536, {}, True, False),                                 # 16 tiles per workgroup
])
def test_update_kernel_one_net_per_workgroup_equals_both_nets_bit_for_bit(case):
    """ppo_update_split_kernel (what every job runs: staging in two round trips, per-layer slab stores on the last tile, the critic's
    phases reading LDS ahead, round 5) against ppo_update_kernel (both nets in one workgroup, the plain loops): equal gradient slabs
    and loss statistics, bit for bit -- every sum keeps its order whatever the kernel waits for."""
    import ctypes

    from tianshou_marl_amd import _abi
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

    D, A, n, M, kw, use_image, vclip = case
    lib = _abi.load()
    lib.tsm_debug_set_update_variant.argtypes = [ctypes.c_int]
    cfg = ops.make_ppo_cfg(**kw)
    out = []
    try:
        for variant in (0, 1):
            lib.tsm_debug_set_update_variant(variant)
            torch.manual_seed(0)
            net = DiscreteActorCritic(D, A, 64, device=DEV, seed=0)
            obs = torch.randn(n, D, device=DEV)
            act = torch.randint(0, A, (n,), dtype=torch.int32, device=DEV)
            logp, adv, ret, v_old = (torch.randn(n, device=DEV) for _ in range(4))
            logp = logp * 0.3 - 1.5
            perm = torch.randperm(n, device=DEV)[:M].contiguous()
            stats = ops.ppo_adv_stats(adv, torch.tensor([0, M], device=DEV), perm=perm)
            slabs, sc = ops.ppo_update_fused(net.flat.data, obs, act, logp, adv, ret, cfg, A, 64, adv_stats=stats[0], perm=perm, M=M,
                                             v_s_old=v_old if vclip else None, image=net.image if use_image else None)
            torch.cuda.synchronize()
            out.append((slabs.clone(), sc.clone()))
    finally:
        lib.tsm_debug_set_update_variant(0)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_stream_ptr_is_torchs_current_stream():
    """_abi.stream_ptr asks torch's C layer for the raw stream: it must be the hipStream_t of torch.cuda.current_stream() on the
    default stream, on a side stream and on the stream a graph is being captured on."""
    assert ops.stream_ptr() == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert ops.stream_ptr() == side.cuda_stream == torch.cuda.current_stream().cuda_stream
    assert ops.stream_ptr() == torch.cuda.current_stream().cuda_stream
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    g = torch.cuda.CUDAGraph()
    seen = []
    with ops.graph_capture(g):
        seen.append((ops.stream_ptr(), torch.cuda.current_stream().cuda_stream))
        ops.call("tsm_u64_add", ops.ptr(ctr), 3, ops.stream_ptr())
    assert seen[0][0] == seen[0][1]
    g.replay(); g.replay()
    torch.cuda.synchronize()
    assert int(ctr.item()) == 6
