"""GPU integration tests (`-m gpu`): batched env kernel vs the numpy MPE oracle, device Collector ->
DeviceVectorReplayBuffer -> PPO.update against a plain-PyTorch (f64, CPU) replica of the reference's
update loop driven by the same permutations."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data.batch import Batch, split_bounds
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

DEV = "cuda"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


@pytest.mark.parametrize("n_env,N", [(5, 3), (64, 3), (130, 8), (7, 1), (9, 2)])
def test_mpe_step_matches_numpy_oracle(n_env, N):
    import mpe_oracle

    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=12, device=DEV, seed=3, auto_reset=False)
    obs0 = env.reset_device().cpu().numpy()
    apos, avel, lpos = (x.cpu().numpy() for x in (env.agent_pos, env.agent_vel, env.landmark_pos))
    assert np.all(np.abs(apos) <= 1) and np.all(np.abs(lpos) <= 1) and np.all(avel == 0)
    assert len(np.unique(apos.round(6))) > n_env  # envs differ
    worlds = [mpe_oracle.SimpleSpreadWorld(N, 12, 0.5) for _ in range(n_env)]
    for e, w in enumerate(worlds):
        w.set_state(apos[e], avel[e], lpos[e])
        np.testing.assert_allclose(obs0[e], w.observe(), rtol=1e-6, atol=1e-6)
    rng = np.random.default_rng(0)
    for step in range(14):
        act = rng.integers(0, 5, (n_env, N)).astype(np.int32)
        obs_next, rew, term, trunc, done = env.step_device(torch.from_numpy(act).to(DEV))
        obs_next, rew, trunc = obs_next.cpu().numpy(), rew.cpu().numpy(), trunc.cpu().numpy()
        for e, w in enumerate(worlds):
            o, r, te, tr = w.step(act[e])
            # f32 kernel vs f64 numpy over up to 14 chained steps (contact forces amplify rounding): 2e-3 abs
            np.testing.assert_allclose(obs_next[e], o, rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(rew[e], r, rtol=2e-3, atol=2e-3)
            assert np.array_equal(trunc[e].astype(bool), tr)
        assert not term.any()
        assert np.array_equal(done.cpu().numpy().astype(bool), trunc[:, 0].astype(bool))
        # without auto-reset the policy input equals obs_next
        assert torch.equal(env.obs_cur, env.obs_next)


@pytest.mark.parametrize("n_env,N", [(64, 3), (40, 8), (16, 1)])
def test_mpe_single_steps_match_oracle_tightly(n_env, N):
    """One step at a time from the SAME state (the oracle world is re-synchronised to the device state before every step, so
    f32 rounding cannot accumulate through the contact dynamics): observations and rewards of the HIP kernel against the
    numpy f64 restatement of the MPE spec at 2e-5 (f32 arithmetic on O(1) quantities) -- worlds squeezed so that contacts
    happen in most steps."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import mpe_oracle

    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=50, device=DEV, seed=5, auto_reset=False)
    env.reset_device()
    env.agent_pos.mul_(0.3)  # crowd the agents: soft contacts in most worlds
    worlds = [mpe_oracle.SimpleSpreadWorld(N, 50, 0.5) for _ in range(n_env)]
    rng = np.random.default_rng(1)
    contacts = 0
    for step in range(10):
        apos, avel, lpos = (x.cpu().numpy().astype(np.float64) for x in (env.agent_pos, env.agent_vel, env.landmark_pos))
        act = rng.integers(0, 5, (n_env, N)).astype(np.int32)
        obs_next, rew, *_ = env.step_device(torch.from_numpy(act).to(DEV))
        obs_next, rew = obs_next.cpu().numpy(), rew.cpu().numpy()
        for e, w in enumerate(worlds):
            w.set_state(apos[e], avel[e], lpos[e])
            if N > 1:
                d = np.linalg.norm(apos[e][:, None] - apos[e][None], axis=-1) + np.eye(N)
                contacts += int((d < 0.3).any())
            o, r, _, _ = w.step(act[e])
            np.testing.assert_allclose(obs_next[e], o, rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(rew[e], r, rtol=2e-5, atol=2e-5)
    assert N == 1 or contacts > n_env  # the contact force path was exercised


def test_mpe_auto_reset_and_determinism():
    mk = lambda seed: DeviceSimpleSpreadVectorEnv(33, 3, max_cycles=4, device=DEV, seed=seed)  # noqa: E731
    a, b, c = mk(1), mk(1), mk(2)
    oa, ob, oc = a.reset_device().clone(), b.reset_device().clone(), c.reset_device().clone()
    assert torch.equal(oa, ob) and not torch.equal(oa, oc)
    act = torch.zeros(33, 3, dtype=torch.int32, device=DEV)
    for t in range(4):
        _, _, _, trunc, done = a.step_device(act)
    assert done.all() and trunc.all()
    assert (a.steps == 0).all()                       # re-initialised in place
    assert not torch.equal(a.obs_cur, a.obs_next)     # next policy input is the reset observation
    assert torch.all(a.obs_cur[:, :, 0:2] == 0)        # zero velocity after reset
    assert not torch.equal(a.obs_cur, oa)             # a new episode, not a replay of the first one
    # reference-style numpy API (BaseVectorEnv contract)
    obs, info = c.reset()
    assert obs.shape == (33,) and set(obs[0]) == {"observations", "agent_ids", "masks"}
    o, r, te, tr, info = c.step(np.zeros((33, 3), int))
    assert r.shape == (33, 3) and te.shape == (33, 3) and info[5]["env_id"] == 5
    partial, _ = c.reset(env_id=[3, 7])
    assert len(partial) == 2


def _mk_job(n_env=32, N=3, T=25, seed=0, **ppo_kw):
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=seed)
    net = DiscreteActorCritic(env.obs_dim, env.n_act, 64, device=DEV, seed=seed)
    algo = PPO(net=net, seed=seed, **ppo_kw)
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=DEV)
    col = Collector(algo, env, buf)
    col.reset()
    return env, net, algo, buf, col


def test_device_collector_fills_buffer_consistently():
    n_env, N, T = 32, 3, 25
    env, net, algo, buf, col = _mk_job(n_env, N, T)
    with policy_within_training_step(algo):
        st = col.collect(n_step=n_env * T)
    assert st.n_collected_steps == n_env * T and st.n_collected_episodes == n_env
    assert np.all(st.lens == T) and st.returns.shape == (n_env, N)      # vector rewards -> (n_ep, N)
    assert len(buf) == n_env * T
    obs, obs_next = buf.obs_store.cpu().numpy(), buf.obs_next_store.cpu().numpy()
    np.testing.assert_array_equal(obs[1:], obs_next[:-1])               # no resets before the last step
    assert buf.trunc_store[-1].all() and not buf.trunc_store[:-1].any() and not buf.term_store.any()
    np.testing.assert_allclose(st.returns, buf.rew_store.double().sum(0).cpu().numpy(), rtol=1e-6)
    assert np.array_equal(buf.sample_indices(0), np.arange(n_env * T))
    assert len(buf.unfinished_index()) == 0
    # stored policy outputs == recomputed with the same parameters (ppo.py:157-161, a2c.py:121-127)
    rows = buf.obs_store.reshape(-1, env.obs_dim)
    again = ops.policy_forward(net.flat.data, rows, 5, 64, mode="given", act=buf.act_store.reshape(-1))
    assert torch.equal(again["logp"], buf.logp_store.reshape(-1))
    assert torch.equal(again["value"], buf.vs_store.reshape(-1))
    # actions are distributed according to the policy (chi-square over all rows, A = 5)
    lg = again["logits"].double()
    p = torch.softmax(lg, -1).mean(0).cpu().numpy()
    counts = np.bincount(buf.act_store.reshape(-1).cpu().numpy(), minlength=5)
    n = counts.sum()
    assert ((counts - n * p) ** 2 / (n * p)).sum() < 40
    # the reference-style view of the same data
    batch, idx = buf.sample(0)
    assert batch.obs.shape == (n_env * T, N, env.obs_dim) and batch.rew.dtype == np.float64
    e, t = 5, 7
    np.testing.assert_array_equal(batch.obs[e * T + t], obs[t, e])
    # second collect continues episodes (n_step mode keeps env state), buffer wraps around
    with policy_within_training_step(algo):
        st2 = col.collect(n_step=n_env * 3)
    assert st2.n_collected_episodes == 0 and len(buf) == n_env * T


def test_collect_n_episode_device():
    env, net, algo, buf, col = _mk_job(8, 3, 6)
    with policy_within_training_step(algo):
        st = col.collect(n_episode=5)   # fewer episodes than envs: only 5 envs are used (collector.py:817-823)
    assert st.n_collected_episodes == 5 and np.all(st.lens == 6) and st.n_collected_steps == 30
    assert buf.index.lengths.cpu().tolist() == [6, 6, 6, 6, 6, 0, 0, 0]
    with policy_within_training_step(algo):
        st = col.collect(n_episode=11)  # 8 envs finish together, then 3 more (surplus envs dropped)
    assert st.n_collected_episodes == 11 and st.n_collected_steps == 8 * 6 + 3 * 6


def _replica_update(net_params, obs, obs_next, act, rew, term, trunc, T, n_env, N, cfg, batch_size, repeat, lr):
    """Plain PyTorch f64 replica of the reference update loop (marl.py:251-268 + ppo.py:146-224) on lanes."""
    from test_gpu_mlp import torch_ppo_loss

    import oracle as orc

    D, H, A = obs.shape[-1], 64, 5
    p = torch.from_numpy(net_params).double().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=lr)
    flat = lambda x: x.reshape(T * n_env * N, *x.shape[3:])  # noqa: E731
    with torch.no_grad():
        z = np.zeros(T * n_env * N)
        _, _, _, _, _, v_s = torch_ppo_loss(p, D, H, A, flat(obs), flat(act), z.astype(np.float32), z.astype(np.float32),
                                            z.astype(np.float32), z.astype(np.float32), dict(cfg, adv_norm=False))
        lg_v = torch_ppo_loss(p, D, H, A, flat(obs_next), flat(act), z.astype(np.float32), z.astype(np.float32),
                              z.astype(np.float32), z.astype(np.float32), dict(cfg, adv_norm=False))
        v_n = lg_v[5]
        logits = torch_ppo_loss(p, D, H, A, flat(obs), flat(act), z.astype(np.float32), z.astype(np.float32),
                                z.astype(np.float32), z.astype(np.float32), dict(cfg, adv_norm=False))[4]
        logp_old = torch.distributions.Categorical(logits=logits).log_prob(torch.from_numpy(flat(act))).numpy()
    L = n_env * N
    ret, adv = orc.gae_lanes(v_s.numpy().reshape(T, L).astype(np.float32), v_n.numpy().reshape(T, L).astype(np.float32),
                             rew.reshape(T, L), term.reshape(T, L), trunc.reshape(T, L), 0.99, 0.95)
    ret, adv = ret.reshape(-1).astype(np.float32), adv.reshape(-1).astype(np.float32)
    losses = []
    for a in range(N):
        # the agent's rows in the reference's sample(0) order (env-major, time-ordered) as lane ids of the time-major stores
        ids = (np.arange(T)[None, :] * n_env + np.arange(n_env)[:, None]).reshape(-1) * N + a
        n = len(ids)
        for _ in range(repeat):
            perm = ids[np.random.permutation(n)]
            for lo, hi in split_bounds(n, batch_size, True):
                mb = perm[lo:hi]
                loss, *_ = torch_ppo_loss(p, D, H, A, flat(obs)[mb], flat(act)[mb], logp_old[mb].astype(np.float32),
                                          adv[mb], ret[mb], v_s.numpy()[mb].astype(np.float32), cfg)
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(loss.item())
    return p.detach().numpy(), np.array(losses), ret, adv


def test_ppo_update_matches_pytorch_replica():
    n_env, N, T = 16, 3, 25
    env, net, algo, buf, col = _mk_job(n_env, N, T, seed=4, shuffle="numpy", dispatch="per_agent", lr=1e-3)
    with policy_within_training_step(algo):
        col.collect(n_step=n_env * T)
        p0 = net.flat.data.cpu().numpy().copy()
        np.random.seed(99)
        stats = algo.update(buf, 128, 2)
    c = lambda x: x.cpu().numpy()  # noqa: E731
    cfg = dict(eps_clip=0.2, dual_clip=None, value_clip=False, adv_norm=True, vf_coef=0.5, ent_coef=0.01)
    np.random.seed(99)
    p_ref, losses, ret, adv = _replica_update(p0, c(buf.obs_store), c(buf.obs_next_store), c(buf.act_store).astype(np.int64),
                                              c(buf.rew_store), c(buf.term_store), c(buf.trunc_store), T, n_env, N, cfg,
                                              128, 2, 1e-3)
    # 30 sequential Adam steps in f32 (device) vs f64 (replica): 2e-4 absolute on parameters of O(0.1-1)
    np.testing.assert_allclose(net.flat.data.cpu().numpy(), p_ref, rtol=0, atol=2e-4)
    d = stats.get_loss_stats_dict()
    assert set(d) >= {"agent_0/loss", "agent_1/actor_loss", "agent_2/vf_loss", "agent_0/ent_loss"}  # marl.py:51-59
    per_agent = len(losses) // N
    for a in range(N):
        np.testing.assert_allclose(d[f"agent_{a}/loss"], losses[a * per_agent:(a + 1) * per_agent].mean(), rtol=2e-3, atol=2e-4)
    assert stats._agent_id_to_stats["agent_0"].gradient_steps == per_agent == 2 * 3  # 400 rows @128 -> 128,128,144
    with pytest.raises(RuntimeError):  # algorithm_base.py:610-615
        algo.update(buf, 64, 1)


def test_ppo_update_pooled_and_options_run_and_learn():
    """pooled dispatch, value/dual clip, grad clipping, return scaling, recompute_advantage: finite losses,
    parameters move, and the policy improves its own surrogate (loss decreases over repeats)."""
    env, net, algo, buf, col = _mk_job(32, 3, 25, seed=1, dispatch="pooled", shuffle="device", dual_clip=2.0,
                                       value_clip=True, max_grad_norm=0.5, return_scaling=True, recompute_advantage=True)
    with policy_within_training_step(algo):
        col.collect(n_step=32 * 25)
        p0 = net.flat.data.clone()
        st = algo.update(buf, 512, 3)
    assert st.gradient_steps == 3 * len(split_bounds(32 * 25 * 3, 512, True))
    assert np.isfinite(list(st.get_loss_stats_dict().values())).all()
    assert not torch.equal(p0, net.flat.data) and torch.isfinite(net.flat.data).all()
    assert algo.ret_rms.count == 3 * 32 * 25 * 3  # one ret_rms.update per _preprocess_batch call (1 + 2 recomputes)
    sd = algo.state_dict()
    assert "_optimizers" in sd and float(sd["_optimizers"][0]["state"][0]["step"]) == st.gradient_steps
    # MARL-trainer entry point: .learn(batch) on one agent's lane (training_coordinator.py:336)
    b = Batch(obs=np.random.randn(50, 18).astype(np.float32), act=np.random.randint(0, 5, 50),
              rew=np.random.randn(50).astype(np.float32), obs_next=np.random.randn(50, 18).astype(np.float32),
              terminated=np.zeros(50, bool))
    out = algo.learn(b)
    assert set(out) == {"loss", "actor_loss", "vf_loss", "ent_loss"} and np.isfinite(list(out.values())).all()


def test_graph_replay_equals_eager_launches():
    """hipGraph capture of the rollout and of the update must not change a single bit of the results."""
    def run(use_graph):
        env = DeviceSimpleSpreadVectorEnv(64, 3, max_cycles=25, device=DEV, seed=11)
        net = DiscreteActorCritic(env.obs_dim, env.n_act, 64, device=DEV, seed=11)
        algo = PPO(net=net, seed=11, shuffle="numpy", dispatch="per_agent", use_graph=use_graph, max_grad_norm=0.5)
        buf = DeviceVectorReplayBuffer(64 * 25, 64, 3, env.obs_dim, device=DEV)
        col = Collector(algo, env, buf, use_graph=use_graph)
        col.reset()
        np.random.seed(5)
        rets, losses = [], []
        for _ in range(4):  # 1st collect is eager in both; graph capture happens on the 2nd, replay on 3rd/4th
            with policy_within_training_step(algo):
                cs = col.collect(n_step=64 * 25)
                ts = algo.update(buf, 512, 2)
            col.reset_buffer(keep_statistics=True)
            rets.append(cs.returns.copy())
            losses.append(ts.get_loss_stats_dict())
        return net.flat.data.cpu().numpy(), rets, losses, buf.obs_store.cpu().numpy(), algo.opt_step

    p_e, r_e, l_e, o_e, s_e = run(False)
    p_g, r_g, l_g, o_g, s_g = run(True)
    assert s_e == s_g == 4 * 3 * 2 * 3
    assert np.array_equal(o_e, o_g)
    for a, b in zip(r_e, r_g):
        assert np.array_equal(a, b)
    assert np.array_equal(p_e, p_g)
    assert l_e == l_g


def test_wrapped_subbuffers_gae_matches_oracle_and_update_runs(oracle):
    """Sub-buffers that wrapped around (more rows collected than slots): the stored rows are rotated and the episode
    the wrap cut in half starts mid-way.  Returns/advantages of the preprocessing pass must equal the oracle's GAE on
    the time-ordered rows (ragged path of tsm_gae_lanes: env_start / env_len), and update() falls back to explicit
    index lists."""
    n_env, N, T, S = 6, 3, 7, 10
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=2)
    net = DiscreteActorCritic(env.obs_dim, env.n_act, 64, device=DEV, seed=2)
    algo = PPO(net=net, seed=2, shuffle="numpy", use_graph=True)
    buf = DeviceVectorReplayBuffer(n_env * S, n_env, N, env.obs_dim, device=DEV)
    col = Collector(algo, env, buf)
    col.reset()
    with policy_within_training_step(algo):
        col.collect(n_step=n_env * 8)
        col.collect(n_step=n_env * 5)  # 13 rows per env into 10 slots: wrapped, insertion index 3
        assert len(buf) == n_env * S and buf.index.insertion_idx.cpu().tolist() == [3] * n_env
        pb = algo._preprocess_batch(buf)
        v_s = ops.policy_forward(net.flat.data, buf.obs_store.reshape(-1, env.obs_dim), 5, 64, mode="none")["value"].view(S, n_env, N)
        v_n = ops.policy_forward(net.flat.data, buf.obs_next_store.reshape(-1, env.obs_dim), 5, 64, mode="none")["value"].view(S, n_env, N)
        order = (np.arange(S) + 3) % S  # oldest stored row first
        g = lambda x: x.cpu().numpy()[order]  # noqa: E731
        ret_o, adv_o = oracle.gae_lanes(g(v_s).reshape(S, -1), g(v_n).reshape(S, -1), g(buf.rew_store).reshape(S, -1),
                                        g(buf.term_store).reshape(S, -1), g(buf.trunc_store).reshape(S, -1), 0.99, 0.95)
        back = np.argsort(order)  # time order -> slot order
        np.testing.assert_allclose(pb["adv"].view(S, -1).cpu().numpy(), adv_o[back], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(pb["ret"].view(S, -1).cpu().numpy(), ret_o[back], rtol=1e-5, atol=1e-6)
        before = net.flat.data.clone()
        st = algo.update(buf, batch_size=32, repeat=1)
    assert all(np.isfinite(v) for v in st.get_loss_stats_dict().values())
    assert not torch.equal(before, net.flat.data)


def test_nan_in_buffer_raises_malformed_buffer_error():
    from tianshou_marl_amd._abi import MalformedBufferError

    env = DeviceSimpleSpreadVectorEnv(4, 3, max_cycles=5, device=DEV, seed=1)
    algo = PPO(net=DiscreteActorCritic(env.obs_dim, 5, 64, device=DEV, seed=1))
    buf = DeviceVectorReplayBuffer(4 * 5, 4, 3, env.obs_dim, device=DEV)
    col = Collector(algo, env, buf, raise_on_nan_in_buffer=True)
    col.reset()
    with policy_within_training_step(algo):
        col.collect(n_step=8)  # fine
        env.agent_vel[2, 1, 0] = float("nan")  # a poisoned env state propagates into the stored observations
        with pytest.raises(MalformedBufferError, match="NaN"):
            col.collect(n_step=4)


def test_update_grid_is_not_monotone_and_slabs_are_sized_for_the_largest_grid():
    """tsm_ppo_update_grid gives a merged last minibatch (320 tiles, two per workgroup) FEWER workgroups than a regular
    one (256 tiles, one per workgroup), so workspaces must be sized by the largest grid over the minibatches, not by the
    grid of the largest minibatch.  9216 rows per agent with batch_size 4096 split into [4096, 5120] (Batch.split
    merge_last): graph replay and eager launches must both run (ops.ppo_update_fused refuses a slab buffer that is too
    small) and agree bit for bit."""
    assert ops.ppo_update_grid(5120) < ops.ppo_update_grid(4096)
    for M in (1, 15, 16, 17, 4096, 5120, 8192, 16384, 65536, 10**6):
        g = ops.ppo_update_grid(M)
        assert 1 <= g <= -(-M // 16)
    with pytest.raises(ValueError):  # capacity check
        P = torch.zeros(ops.policy_param_count(18, 64, 5), device=DEV)
        z = torch.zeros(4096, device=DEV)
        ops.ppo_update_fused(P, torch.zeros(4096, 18, device=DEV), torch.zeros(4096, dtype=torch.int32, device=DEV), z, z, z,
                             ops.make_ppo_cfg(adv_norm=False), 5, 64, slabs=torch.zeros(8, P.numel(), device=DEV))
    finals = []
    for use_graph in (True, False):
        np.random.seed(5)
        n_env, T = 384, 24
        env = DeviceSimpleSpreadVectorEnv(n_env, 3, max_cycles=T, device=DEV, seed=2)
        net = DiscreteActorCritic(18, 5, 64, device=DEV, seed=1)
        algo = PPO(net=net, lr=1e-3, seed=3, use_graph=use_graph, dispatch="per_agent")
        buf = DeviceVectorReplayBuffer(n_env * T, n_env, 3, 18, device=DEV)
        col = Collector(algo, env, buf)
        col.reset()
        assert [e - s for s, e in split_bounds(n_env * T, 4096)] == [4096, 5120]
        for _ in range(3):
            with policy_within_training_step(algo):
                col.collect(n_step=n_env * T)
                st = algo.update(buf, 4096, 1)
            col.reset_buffer(keep_statistics=True)
        assert np.isfinite(st.get_loss_stats_dict()["agent_0/loss"])
        finals.append(net.flat.data.clone())
    assert torch.equal(finals[0], finals[1])


def test_headline_job_learns_with_a_stable_configuration():
    """Learning regression on the headline workload (simple_spread N=3, 1024 envs, shared PPO, per-agent dispatch, minibatch
    4096) through the product path (persistent rollout + captured update graph): the mean episode return improves from
    the random policy's ~ -26 and everything stays finite over 600 updates.  Configuration: gamma = 0.95 and
    max_grad_norm = 0.5 -- with the reference's defaults (gamma 0.99, no clipping, truncation bootstrapped from the critic)
    the critic diverges after ~600 updates, in this engine AND in an independent PyTorch replica alike
    (tools/learning_curve.py, profiles/r02_learning_curves.txt)."""
    E, N, T = 1024, 3, 25
    env = DeviceSimpleSpreadVectorEnv(E, N, device=DEV, seed=1626)
    net = DiscreteActorCritic(env.obs_dim, 5, 64, device=DEV, seed=1626)
    algo = PPO(net=net, lr=3e-4, gamma=0.95, max_grad_norm=0.5, dispatch="per_agent", shuffle="device", seed=1626)
    buf = DeviceVectorReplayBuffer(E * T, E, N, env.obs_dim, device=DEV)
    col = Collector(algo, env, buf)
    col.reset()
    curve = []
    for i in range(600):
        with policy_within_training_step(algo):
            cs = col.collect(n_step=E * T)
            ts = algo.update(buf, 4096, 1)
        col.reset_buffer(keep_statistics=True)
        if i % 100 == 0 or i == 599:
            d = ts.get_loss_stats_dict()
            assert all(np.isfinite(v) for v in d.values()), (i, d)
            curve.append(float(cs.returns.mean()))
    assert curve[0] < -22.0, curve              # the untrained policy
    assert curve[-1] > -15.0, curve             # measured: -11.3 after 600 updates
    assert min(curve[2:]) > curve[0] + 5.0, curve  # no collapse on the way
    assert torch.isfinite(net.flat.data).all() and torch.isfinite(algo.exp_avg_sq).all()
