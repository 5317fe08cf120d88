"""GPU tests (`-m gpu`) of the batched simple_tag env (csrc/mpe_tag.hip) and the two-team configuration built on it
(BASELINE configs[4]): dynamics vs the numpy float64 oracle (same published spec; parity with pettingzoo itself is
unpinned), the vector-env contract, grouped policies through the Collector, and the self-play / league trainers
learning from the device buffer."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))

if torch.cuda.is_available():
    import mpe_tag_oracle

    from tianshou_marl_amd.algorithm.multiagent import (
        FlexibleMultiAgentPolicyManager,
        LeaguePlayTrainer,
        SelfPlayTrainer,
        agent_batches_from_buffer,
    )
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe_tag import DeviceSimpleTagVectorEnv
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

DEV = "cuda"


@pytest.mark.parametrize("n_env,n_adv,n_good,n_obst", [(33, 3, 1, 2), (8, 2, 2, 1), (5, 1, 1, 0), (64, 4, 2, 3)])
def test_tag_step_matches_numpy_oracle(n_env, n_adv, n_good, n_obst):
    T = 12
    env = DeviceSimpleTagVectorEnv(n_env, n_good, n_adv, n_obst, max_cycles=T, device=DEV, seed=3, auto_reset=False)
    N = env.n_agent
    assert env.obs_dim == 4 + 2 * n_obst + 2 * (N - 1) + 2 * n_good
    assert env.agents[:n_adv] == [f"adversary_{i}" for i in range(n_adv)] and len(env.agents) == N
    obs0 = env.reset_device().clone()
    # squeeze the worlds so that contacts (agent-agent, agent-obstacle) and the boundary penalty actually occur
    env.agent_pos.mul_(0.35)
    env.agent_pos[: n_env // 2, -1] += 0.8
    worlds = []
    for e in range(n_env):
        w = mpe_tag_oracle.SimpleTagWorld(n_adv, n_good, n_obst, max_cycles=T)
        w.set_state(env.agent_pos[e].cpu().numpy(), env.agent_vel[e].cpu().numpy(), env.landmark_pos[e, :n_obst].cpu().numpy())
        worlds.append(w)
    assert obs0.shape == (n_env, N, env.obs_dim)
    rng = np.random.default_rng(0)
    saw_contact = saw_bound = False
    for t in range(T):
        act = rng.integers(0, 5, (n_env, N))
        obs_next, rew, term, trunc, done = env.step_device(torch.as_tensor(act, dtype=torch.int32, device=DEV))
        exp = [w.step(act[e]) for e, w in enumerate(worlds)]
        exp_obs = np.stack([x[0] for x in exp])
        exp_rew = np.stack([x[1] for x in exp])
        np.testing.assert_allclose(obs_next.cpu().numpy(), exp_obs, rtol=2e-4, atol=2e-5)
        got_rew = rew.cpu().numpy()
        # contact / boundary indicators are discontinuous: compare where the float64 world is not on an edge
        close = np.abs(got_rew - exp_rew) < 1e-3
        assert close.mean() > 0.995, (t, close.mean())
        saw_contact |= bool((exp_rew[:, :n_adv] > 0).any())
        saw_bound |= bool(((exp_rew[:, n_adv:] < 0) & (exp_rew[:, n_adv:] > -10)).any())
        assert not term.any() and bool(trunc.all()) == (t == T - 1) and bool(done.all()) == (t == T - 1)
        # adversaries share one reward; padding of the good agents' rows stays zero
        assert np.allclose(got_rew[:, :n_adv], got_rew[:, :1])
        assert (obs_next[:, n_adv:, env.obs_dim - 2 * n_good + 2 * (n_good - 1):] == 0).all()
        np.testing.assert_allclose(env.obs_cur.cpu().numpy(), exp_obs, rtol=2e-4, atol=2e-5)  # no auto reset here
    assert saw_contact and saw_bound


def test_tag_auto_reset_determinism_and_host_api():
    def run(seed):
        env = DeviceSimpleTagVectorEnv(16, device=DEV, seed=seed, max_cycles=4)
        env.reset_device()
        outs = []
        for t in range(9):
            act = torch.full((16, env.n_agent), t % 5, dtype=torch.int32, device=DEV)
            obs_next, rew, term, trunc, done = env.step_device(act)
            outs.append((obs_next.clone(), env.obs_cur.clone(), done.clone(), env.steps.clone()))
        return env, outs

    env, a = run(1)
    _, b = run(1)
    _, c = run(2)
    for x, y in zip(a, b):
        assert all(torch.equal(p, q) for p, q in zip(x, y))
    assert not torch.equal(a[0][0], c[0][0])
    for t, (obs_next, obs_cur, done, steps) in enumerate(a):
        finished = (t + 1) % 4 == 0
        assert bool(done.all()) == finished
        if finished:  # the policy's next input is the reset observation (zero velocities), not the terminal one
            assert not torch.equal(obs_next, obs_cur) and (obs_cur[:, :, :2] == 0).all() and (steps == 0).all()
        else:
            assert torch.equal(obs_next, obs_cur)
    assert int(env.episode_ctr[0]) == 3  # initial reset + two auto resets
    # numpy-style vector-env contract (parallel-mode dict layout)
    obs, info = env.reset()
    assert obs.shape == (16,) and set(obs[0]["observations"]) == set(env.agents) and info[3]["env_id"] == 3
    o, r, te, tr, info = env.step(np.zeros((16, env.n_agent), int))
    assert r.shape == (16, env.n_agent) and te.dtype == bool and o[0]["agent_ids"] == env.agents
    with pytest.raises(ValueError):
        DeviceSimpleTagVectorEnv(4, num_good=5, num_adversaries=5, device=DEV)


@pytest.mark.parametrize("n_adv,n_good,n_obst", [(3, 1, 2), (1, 1, 4), (5, 3, 0)])
def test_tag_auto_reset_draws_the_same_worlds_as_the_reset_kernel(n_adv, n_good, n_obst):
    """The step kernel re-initialises finished envs lane-wise (agent lane i draws agent i and obstacles i, i + NA, ...);
    the stand-alone reset kernel does it one thread per env.  Same Philox counters => the same worlds, bit for bit."""
    E, T = 21, 3
    a = DeviceSimpleTagVectorEnv(E, n_good, n_adv, n_obst, max_cycles=T, device=DEV, seed=9)
    b = DeviceSimpleTagVectorEnv(E, n_good, n_adv, n_obst, max_cycles=T, device=DEV, seed=9)
    a.reset_device()
    act = torch.zeros(E, a.n_agent, dtype=torch.int32, device=DEV)
    for _ in range(T):
        a.step_device(act)  # the last step truncates and auto-resets (episode 1)
    b.reset_device()
    obs_b = b.reset_device().clone()  # second reset: episode 1
    assert torch.equal(a.agent_pos, b.agent_pos) and torch.equal(a.agent_vel, b.agent_vel)
    if n_obst:
        assert torch.equal(a.landmark_pos[:, :n_obst], b.landmark_pos[:, :n_obst])
    assert torch.equal(a.obs_cur, obs_b)


def _team_policies(env, seed=0, **kw):
    mk = lambda s: PPO(net=DiscreteActorCritic(env.obs_dim, 5, 64, device=DEV, seed=s), seed=s, use_graph=False, **kw)  # noqa: E731
    return {"adversaries": mk(seed), "good": mk(seed + 1)}


def test_team_members_are_served_by_one_forward_call_with_the_same_draws():
    """Consecutive agent columns that share a policy go through ONE fused forward on agent-major rows; counters (and
    therefore sampled actions, log-probs, values) are identical to column-by-column calls."""
    E = 37
    env = DeviceSimpleTagVectorEnv(E, device=DEV, seed=1)
    teams = _team_policies(env)
    mgr = FlexibleMultiAgentPolicyManager(teams, env, mode="grouped", agent_groups=env.agent_groups)
    assert [(a, k) for a, k, _ in mgr._policy_runs()] == [(0, env.n_adv), (env.n_adv, env.n_good)]
    obs = torch.randn(E, env.n_agent, env.obs_dim, device=DEV)
    tick = torch.tensor([12345], dtype=torch.int64, device=DEV)
    with policy_within_training_step(mgr):
        got = mgr.act_device(obs, offset_dev=tick)
        for a, name in enumerate(env.agents):
            ref = mgr.policy_map[name].act_device(obs[:, a].contiguous(), offset_dev=tick, row_offset=a * E)
            for f in ("act", "logp", "value"):
                assert torch.equal(got[f].view(E, env.n_agent)[:, a], ref[f].view(E)), (name, f)
    # a map with the same policy on NON-adjacent columns falls back to separate calls for them
    mgr.policy_map["adversary_1"] = teams["good"]
    assert [(a, k) for a, k, _ in mgr._policy_runs()] == [(0, 1), (1, 1), (2, 1), (3, 1)]


def test_grouped_policies_collect_and_train_on_tag():
    n_env, T = 64, 25
    env = DeviceSimpleTagVectorEnv(n_env, device=DEV, seed=5, max_cycles=T)
    teams = _team_policies(env)
    mgr = FlexibleMultiAgentPolicyManager(teams, env, mode="grouped", agent_groups=env.agent_groups)
    assert mgr.policy_map["adversary_2"] is teams["adversaries"] and mgr.policy_map["agent_0"] is teams["good"]
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, env.n_agent, env.obs_dim, device=DEV)
    col = Collector(mgr, env, buf)
    col.reset()
    with policy_within_training_step(mgr):
        st = col.collect(n_step=n_env * T)
        st2 = col.collect(n_step=n_env * T) if False else None
    assert st.n_collected_episodes == n_env and st.returns.shape == (n_env, env.n_agent) and st2 is None
    # team rewards: adversaries share theirs; the prey's is never positive
    assert np.allclose(st.returns[:, :env.n_adv], st.returns[:, :1]) and (st.returns[:, env.n_adv:] <= 1e-6).all()
    # every stored action was drawn from the policy of the agent's TEAM
    from tianshou_marl_amd import ops

    for a, name in enumerate(env.agents):
        pol = mgr.policy_map[name]
        rows = buf.obs_store[:T, :, a].reshape(-1, env.obs_dim)
        ref = ops.policy_forward(pol.net.flat.data, rows, 5, 64, mode="given", act=buf.act_store[:T, :, a].reshape(-1))
        assert torch.allclose(ref["logp"], buf.logp_store[:T, :, a].reshape(-1), rtol=1e-5, atol=1e-6), name
    batch = agent_batches_from_buffer(buf, env.agents)
    # self-play: only the prey learns; snapshots of it join the opponent pool
    sp = SelfPlayTrainer(mgr, main_agent_id="good", snapshot_interval=2, opponent_pool_size=3)
    team_batch = batch  # trainers index by policy key: give each team its (first member's) batch under the team name
    team_batch["good"], team_batch["adversaries"] = batch["agent_0"], batch["adversary_0"]
    before = {k: p.net.flat.data.clone() for k, p in teams.items()}
    for _ in range(4):
        out = sp.train_step(team_batch)
        assert list(out) == ["good"] and np.isfinite(out["good"]["loss"])
    assert not torch.equal(before["good"], teams["good"].net.flat.data)
    assert torch.equal(before["adversaries"], teams["adversaries"].net.flat.data)
    assert len(sp.opponent_pool) == 2 and sp._sample_opponent() is not teams["good"]
    # a snapshot acts like the policy it was copied from (deep copy of flat parameters and their LDS image)
    snap = sp.opponent_pool[-1]
    obs = torch.randn(32, env.obs_dim, device=DEV)
    assert torch.equal(ops.policy_forward(snap.net.flat.data, obs, 5, 64, image=snap.net.image, mode="mode")["act"],
                       ops.policy_forward(teams["good"].net.flat.data, obs, 5, 64, mode="mode")["act"])
    # league: two matched teams learn per step, ratings move with results
    lg = LeaguePlayTrainer(mgr, matchmaking="elo", games_per_evaluation=2)
    out = lg.train_step(team_batch)
    assert set(out) == {"adversaries", "good"}
    lg.update_match_result("adversaries", "good")
    assert lg.elo_ratings["adversaries"] > 1000 > lg.elo_ratings["good"]


def test_configs4_at_full_size_4096_envs():
    """BASELINE configs[4] at its stated per-job size: 4096 simple_tag worlds (3 adversaries v 1 prey, 2 obstacles),
    grouped policies, one collect of 25 vector steps, then one self-play and one league training step on the device
    rows.  The env kernel is spot-checked against the numpy oracle on a 64-env slice of the SAME 4096-env launches."""
    n_env, T, chk = 4096, 25, 64
    env = DeviceSimpleTagVectorEnv(n_env, device=DEV, seed=7, max_cycles=T, auto_reset=False)
    N = env.n_agent
    env.reset_device()
    env.agent_pos.mul_(0.5)  # denser worlds: contacts inside the checked slice
    worlds = []
    for e in range(chk):
        w = mpe_tag_oracle.SimpleTagWorld(env.n_adv, env.n_good, env.n_obst, max_cycles=T)
        w.set_state(env.agent_pos[e].cpu().numpy(), env.agent_vel[e].cpu().numpy(),
                    env.landmark_pos[e, :env.n_obst].cpu().numpy())
        worlds.append(w)
    rng = np.random.default_rng(1)
    for t in range(6):
        act = rng.integers(0, 5, (n_env, N))
        obs_next, rew, term, trunc, done = env.step_device(torch.as_tensor(act, dtype=torch.int32, device=DEV))
        exp = [w.step(act[e]) for e, w in enumerate(worlds)]
        np.testing.assert_allclose(obs_next[:chk].cpu().numpy(), np.stack([x[0] for x in exp]), rtol=2e-4, atol=2e-5)
        close = np.abs(rew[:chk].cpu().numpy() - np.stack([x[1] for x in exp])) < 1e-3
        assert close.mean() > 0.99
        assert torch.isfinite(obs_next).all() and torch.isfinite(rew).all()
    # the training job at this size
    env = DeviceSimpleTagVectorEnv(n_env, device=DEV, seed=5, max_cycles=T)
    teams = _team_policies(env)
    mgr = FlexibleMultiAgentPolicyManager(teams, env, mode="grouped", agent_groups=env.agent_groups)
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=DEV)
    col = Collector(mgr, env, buf)
    col.reset()
    with policy_within_training_step(mgr):
        st = col.collect(n_step=n_env * T)
    assert st.n_collected_steps == n_env * T and st.n_collected_episodes == n_env
    assert st.returns.shape == (n_env, N) and np.isfinite(st.returns).all()
    assert np.allclose(st.returns[:, :env.n_adv], st.returns[:, :1]) and (st.returns[:, env.n_adv:] <= 1e-6).all()
    assert len(buf) == n_env * T and not buf.hasnull()
    batch = agent_batches_from_buffer(buf, env.agents)
    batch["good"], batch["adversaries"] = batch["agent_0"], batch["adversary_0"]
    assert len(batch["good"].obs) == n_env * T
    before = {k: p.net.flat.data.clone() for k, p in teams.items()}
    sp = SelfPlayTrainer(mgr, main_agent_id="good", snapshot_interval=1, opponent_pool_size=2)
    out = sp.train_step(batch)
    assert list(out) == ["good"] and np.isfinite(out["good"]["loss"])
    assert not torch.equal(before["good"], teams["good"].net.flat.data)
    assert torch.equal(before["adversaries"], teams["adversaries"].net.flat.data) and len(sp.opponent_pool) == 1
    lg = LeaguePlayTrainer(mgr, matchmaking="elo", games_per_evaluation=2)
    out = lg.train_step(batch)
    assert set(out) == {"adversaries", "good"} and all(np.isfinite(v["loss"]) for v in out.values())
    assert not torch.equal(before["adversaries"], teams["adversaries"].net.flat.data)
    for p in teams.values():
        assert torch.isfinite(p.net.flat.data).all()


def _tag_job(n_env, n_adv, n_good, n_obst, T, fused, slots, shared=False, deterministic=False):
    env = DeviceSimpleTagVectorEnv(n_env, n_good, n_adv, n_obst, max_cycles=T, device=DEV, seed=11)
    mk = lambda s: PPO(net=DiscreteActorCritic(env.obs_dim, 5, 64, device=DEV, seed=s), seed=s, deterministic_eval=deterministic)  # noqa: E731
    if shared:
        mgr = FlexibleMultiAgentPolicyManager(mk(21), env, mode="shared")
    else:
        mgr = FlexibleMultiAgentPolicyManager({"adversaries": mk(21), "good": mk(22)}, env, mode="grouped",
                                              agent_groups=env.agent_groups)
    buf = DeviceVectorReplayBuffer(n_env * slots, n_env, env.n_agent, env.obs_dim, device=DEV)
    col = Collector(mgr, env, buf, fused_rollout=fused, use_graph=False)
    col.reset()
    return env, mgr, buf, col


# (512, 3, 1, 2): the configs[4] shard (128 workgroups); (300, 2, 2, 1): two good agents; (9, 1, 1, 0): no obstacles, a
# ragged last workgroup; (40, 4, 2, 3): six agents -> two worlds per workgroup, four dead tile rows; (14, 3, 2, 4): 15 rows x 9
# entities = 135 pair force tasks -> the second pass of the two waves that evaluate them beside the head, and a ragged last workgroup
@pytest.mark.parametrize("n_env,n_adv,n_good,n_obst,T,steps,shared", [
    (512, 3, 1, 2, 25, 25, False), (300, 2, 2, 1, 5, 12, False), (9, 1, 1, 0, 4, 9, False), (40, 4, 2, 3, 6, 7, False),
    (14, 3, 2, 4, 5, 8, False), (64, 3, 1, 2, 7, 10, True)])
def test_fused_tag_rollout_is_bit_identical_to_unfused(n_env, n_adv, n_good, n_obst, T, steps, shared):
    """csrc/rollout_tag.hip: ONE launch per collect() must reproduce, bit for bit, the unfused Collector loop under grouped
    policies (per team policy_forward on agent-major rows -> tag step -> buffer add), across episode ends / re-initialised
    worlds and across two consecutive collects."""
    slots = steps + 3 + 1
    outs = []
    for fused in (False, True):
        env, mgr, buf, col = _tag_job(n_env, n_adv, n_good, n_obst, T, fused, slots, shared=shared)
        assert col._can_fuse() == fused
        with policy_within_training_step(mgr):
            st1 = col.collect(n_step=n_env * steps)
            st2 = col.collect(n_step=n_env * 3)  # continues mid-episode
        outs.append(dict(
            obs=buf.obs_store.clone(), obs_next=buf.obs_next_store.clone(), act=buf.act_store.clone(),
            rew=buf.rew_store.clone(), trunc=buf.trunc_store.clone(), term=buf.term_store.clone(),
            logp=buf.logp_store.clone(), vs=buf.vs_store.clone(), done=buf.done_store.clone(),
            state=buf.index.state.clone(), apos=env.agent_pos.clone(), avel=env.agent_vel.clone(),
            lpos=env.landmark_pos.clone(), steps=env.steps.clone(), ep=env.episode_ctr.clone(), obs_cur=env.obs_cur.clone(),
            tick=env.rng_tick.clone(), ret1=st1.returns, len1=st1.lens, n1=st1.n_collected_episodes,
            ret2=st2.returns, n2=st2.n_collected_episodes))
        if fused:
            # V(obs_next) as the kernel stores it (next step's V(obs) / the terminal observation's own pass / the pass behind the
            # loop) == the row's own team's critic on the stored obs_next rows, bit for bit (a2c.py:124)
            from tianshou_marl_amd import ops

            n_rows, N = steps + 3, env.n_agent
            for ai in range(N):
                pol = mgr.policy_map[env.agents[ai]]
                rows = buf.obs_next_store[:n_rows, :, ai].reshape(-1, env.obs_dim).contiguous()
                ref = ops.policy_forward(pol.net.flat.data, rows, 5, 64, mode="none")["value"].reshape(n_rows, n_env)
                assert torch.equal(buf.vnext_store[:n_rows, :, ai], ref), ai
            cols = buf.column_outputs
            assert isinstance(cols, list) and len(cols) == N and cols[0][0] == id(mgr.policy_map[env.agents[0]])
    a, b = outs
    assert a["n1"] == n_env * (steps // T) and a["n1"] + a["n2"] == n_env * ((steps + 3) // T)
    for k in a:
        if isinstance(a[k], torch.Tensor):
            assert torch.equal(a[k], b[k]), k
        elif isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


def test_fused_tag_rollout_takes_the_mode_outside_training_steps():
    """deterministic_eval policies act with dist.mode outside a training step (reinforce.py:167-192): the fused kernel
    must switch per policy exactly as the unfused calls do."""
    outs = []
    for fused in (False, True):
        env, mgr, buf, col = _tag_job(32, 3, 1, 2, 6, fused, 12, deterministic=True)
        col.collect(n_step=32 * 8)  # not within a training step: argmax actions
        outs.append((buf.act_store.clone(), buf.logp_store.clone(), buf.obs_store.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize("shared", [False, True])
def test_learn_on_stored_rollout_outputs_equals_learn_that_recomputes(shared):
    """the agent batches of a fused tag rollout carry each column's stored logp_old / V(obs) / V(obs_next),
    tagged with the policy and parameter version that produced them; `PPO.learn` takes them when it IS that policy at that version
    (no policy_forward launches) and recomputes otherwise -- a shared policy's second learn() of a step has moved on, as in the
    reference, where every learn() evaluates its own rows afresh (ppo.py:157-161).  Losses and parameters after two training steps:
    identical to learners that always recompute; graph and eager alike."""
    from tianshou_marl_amd import ops

    res = []
    # (reuse stored outputs, learn() as a graph replay, agent batches as env-major copies / as handles on the stores)
    for reuse, graph, copies in ((True, True, True), (False, True, True), (True, False, True), (True, True, False), (True, False, False)):
        env, mgr, buf, col = _tag_job(64, 3, 1, 2, 25, True, 25, shared=shared)
        pols = {id(p): p for p in mgr.policy_map.values()}
        for p in pols.values():
            p.reuse_rollout_outputs, p.use_graph, p.shuffle = reuse, graph, "device"
        losses = []
        launches = []
        orig = ops.policy_forward
        for step in range(3):
            with policy_within_training_step(mgr):
                col.collect(n_step=64 * 25)
                batch = agent_batches_from_buffer(buf, env.agents, global_state=False, copies=copies)
                if copies:
                    assert "v_next" in batch["agent_0"] and "outputs_version" in batch["adversary_0"]
                else:
                    assert "obs" not in batch["agent_0"] and batch["agent_0"].store_rows.store.column_outputs is not None
                count = [0]

                def counting(*a, **k):
                    count[0] += 1
                    return orig(*a, **k)

                ops.policy_forward = counting
                try:
                    for name in ("adversary_0", "adversary_1", "agent_0"):
                        losses.append(dict(mgr.policy_map[name].learn(batch[name], 800, 1)))
                finally:
                    ops.policy_forward = orig
                launches.append(count[0])
            col.reset_buffer(keep_statistics=True)
        torch.cuda.synchronize()
        res.append((losses, [p.net.flat.data.clone() for p in pols.values()], launches))
    (l0, p0, n0), (l1, p1, n1), (l2, p2, n2), (l3, p3, n3), (l4, p4, n4) = res
    assert l0 == l1 == l2 == l3 == l4
    for x, *rest in zip(p0, p1, p2, p3, p4):
        assert all(torch.equal(x, y) for y in rest)
    assert n4 == n2
    # eager learners (the third run) show what was launched: with stored outputs only a policy that has ALREADY learned this step
    # recomputes -- grouped: adversary_1 (its policy moved at adversary_0's call): 2 forwards; one shared policy: calls 2 and 3: 4
    assert n2 == ([4, 4, 4] if shared else [2, 2, 2])


def test_agent_batches_fast_path_equals_the_indexed_gather():
    """agent_batches_from_buffer reads equally filled sub-buffers as strided views; the general path gathers by
    sample_indices(0).  Same rows, same order, same dtypes; `only` restricts the agents built."""
    env, mgr, buf, col = _tag_job(48, 3, 1, 2, 6, True, 10)
    with policy_within_training_step(mgr):
        col.collect(n_step=48 * 10)
    assert buf.host_uniform_len() == 10
    fast = agent_batches_from_buffer(buf, env.agents)
    some = agent_batches_from_buffer(buf, env.agents, only=["agent_0"])
    assert "agent_0" in some and "adversary_0" not in some
    buf._host_rows = None  # fill level unknown on the host: the indexed path
    slow = agent_batches_from_buffer(buf, env.agents)
    for name in env.agents:
        for k in ("obs", "act", "rew", "obs_next", "terminated", "truncated"):
            assert fast[name][k].dtype == slow[name][k].dtype and torch.equal(fast[name][k], slow[name][k]), (name, k)
    assert torch.equal(fast["global_obs"], slow["global_obs"]) and torch.equal(fast["global_obs_next"], slow["global_obs_next"])
    assert torch.equal(some["agent_0"].obs, slow["agent_0"].obs)


def test_agent_batches_of_a_buffer_without_an_obs_next_store():
    """a uniformly filled `ignore_obs_next=True` buffer has no obs_next store -- agent_batches_from_buffer must take
    the indexed gather (obs at next(index), buffer_base.py:612-616), not dereference the missing store."""
    env = DeviceSimpleTagVectorEnv(48, 1, 3, 2, max_cycles=6, device=DEV, seed=11)
    mk = lambda s: PPO(net=DiscreteActorCritic(env.obs_dim, 5, 64, device=DEV, seed=s), seed=s)  # noqa: E731
    mgr = FlexibleMultiAgentPolicyManager({"adversaries": mk(21), "good": mk(22)}, env, mode="grouped", agent_groups=env.agent_groups)
    buf = DeviceVectorReplayBuffer(48 * 10, 48, env.n_agent, env.obs_dim, device=DEV, ignore_obs_next=True)
    col = Collector(mgr, env, buf, use_graph=False, fused_rollout=False)
    col.reset()
    with policy_within_training_step(mgr):
        col.collect(n_step=48 * 10)
    assert buf.host_uniform_len() == 10 and buf.obs_next_store is None
    got = agent_batches_from_buffer(buf, env.agents)
    idx = buf.index.sample_indices_all()
    ref = buf.get_device(idx.cpu().numpy())
    for a, name in enumerate(env.agents):
        assert torch.equal(got[name].obs, ref["obs"][:, a]) and torch.equal(got[name].obs_next, ref["obs_next"][:, a])
        assert got[name].act.dtype == torch.int64 and torch.equal(got[name].rew, ref["rew"][:, a])
    assert torch.equal(got["global_obs_next"], ref["obs_next"].reshape(len(idx), -1))


@pytest.mark.parametrize("n_env,n_adv,n_good,n_obst", [(48, 3, 1, 2), (24, 4, 2, 3)])
def test_tag_single_steps_match_oracle_tightly(n_env, n_adv, n_good, n_obst):
    """One step at a time from the SAME state (the oracle world is re-synchronised to the device state before every step):
    observations and rewards of the HIP kernel against the numpy f64 restatement of the MPE spec at 2e-5 / exact catches."""
    env = DeviceSimpleTagVectorEnv(n_env, n_good, n_adv, n_obst, max_cycles=50, device=DEV, seed=5, auto_reset=False)
    N = env.n_agent
    env.reset_device()
    env.agent_pos.mul_(0.3)
    worlds = [mpe_tag_oracle.SimpleTagWorld(n_adv, n_good, n_obst, max_cycles=50) for _ in range(n_env)]
    rng = np.random.default_rng(2)
    caught = 0
    for step in range(10):
        apos, avel = env.agent_pos.cpu().numpy().astype(np.float64), env.agent_vel.cpu().numpy().astype(np.float64)
        lpos = env.landmark_pos[:, :n_obst].cpu().numpy().astype(np.float64)
        act = rng.integers(0, 5, (n_env, N))
        obs_next, rew, *_ = env.step_device(torch.as_tensor(act, dtype=torch.int32, device=DEV))
        obs_next, rew = obs_next.cpu().numpy(), rew.cpu().numpy()
        for e, w in enumerate(worlds):
            w.set_state(apos[e], avel[e], lpos[e])
            o, r = w.step(act[e])[:2]
            np.testing.assert_allclose(obs_next[e], o, rtol=2e-5, atol=2e-5)
            near_threshold = np.abs(np.asarray(r) - rew[e]) > 5.0  # a contact decided the other way within f32 rounding
            if not near_threshold.any():
                np.testing.assert_allclose(rew[e], r, rtol=2e-5, atol=2e-4)
            caught += int(rew[e, 0] >= 10.0)
    assert caught > 0  # adversaries did catch the prey in some worlds: the reward path was exercised
