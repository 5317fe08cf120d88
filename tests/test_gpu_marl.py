"""GPU tests (`-m gpu`) of the multi-agent host layer on top of the HIP path:
MultiAgentPolicy / FlexibleMultiAgentPolicyManager dispatch vs the reference fixture, per-agent (independent)
algorithms through MARLDispatcher, the trainers' `.learn()` on device batches, and the reference-style HOST
collect loop (DummyVectorEnv of parallel-mode envs = BASELINE configs[0]) feeding the device buffer."""
import os
import warnings

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.algorithm.multiagent import (
        FlexibleMultiAgentPolicyManager,
        MultiAgentOnPolicyAlgorithm,
        MultiAgentPolicy,
        SimultaneousTrainer,
        agent_batches_from_buffer,
    )
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data import Batch
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env import DummyVectorEnv, EnhancedPettingZooEnv
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.env.spaces import Box, Discrete
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


class Env:
    def __init__(self, n):
        self.agents = [f"agent_{i}" for i in range(n)]
        self.agent_idx = {a: i for i, a in enumerate(self.agents)}


class ArgmaxPolicy(torch.nn.Module if torch.cuda.is_available() else object):
    """act = argmax(obs @ W): the deterministic mock the fixture was generated with (make_fixtures.py)."""

    def __init__(self, W):
        super().__init__()
        self.W = torch.as_tensor(W).to(DEV)
        self.device = torch.device(DEV)
        self.calls = 0

    def forward(self, batch, state=None, **kw):
        self.calls += 1
        obs = batch.obs.obs if isinstance(batch.obs, Batch) else batch.obs
        obs = torch.as_tensor(obs).to(DEV, torch.float32)
        logits = obs @ self.W
        return Batch(act=logits.argmax(-1), state=None, logits=logits)


def test_multiagent_policy_dispatch_matches_reference_fixture():
    g = np.load(os.path.join(GOLD, "marl_dispatch.npz"))
    env = Env(3)
    agent_id = np.array([env.agents[i] for i in g["agent_rows"]], dtype=object)
    batch = Batch(obs=Batch(agent_id=agent_id, obs=g["obs"]), info=Batch())
    pols = {a: ArgmaxPolicy(g["Ws"][i]) for i, a in enumerate(env.agents)}
    res = MultiAgentPolicy(pols, env.agent_idx)(batch)
    assert np.array_equal(res.act.cpu().numpy(), g["act_independent"])  # bit-exact scatter
    assert set(res.out.get_keys()) == set(env.agents) and all(p.calls == 1 for p in pols.values())
    # shared: ONE forward over all rows (flexible_policy.py:202-231)
    shared = ArgmaxPolicy(g["Ws"][0])
    res = FlexibleMultiAgentPolicyManager(shared, env, mode="shared")(batch)
    assert shared.calls == int(g["shared_calls"]) == 1
    assert np.array_equal(res.act.cpu().numpy(), g["act_shared"])
    # grouped: policy_map semantics (the reference's own grouped forward yields no act: quirk Q6)
    assert int(g["grouped_forward_has_act"]) == 0
    pa, pb = ArgmaxPolicy(g["Ws"][1]), ArgmaxPolicy(g["Ws"][2])
    mgr = FlexibleMultiAgentPolicyManager({"g0": pa, "g1": pb}, env, mode="grouped",
                                          agent_groups={"g0": ["agent_0", "agent_1"], "g1": ["agent_2"]})
    res = mgr(batch)
    assert np.array_equal(res.act.cpu().numpy(), g["act_grouped_policy_map"])
    # an agent with no rows gets empty out/state entries and leaves act untouched (marl.py:149-152)
    only0 = Batch(obs=Batch(agent_id=np.array(["agent_0"] * 4, dtype=object), obs=g["obs"][:4]), info=Batch())
    res = MultiAgentPolicy(pols, env.agent_idx)(only0)
    assert res.out.agent_1.is_empty() and res.act.shape == (4,)


def _ppo(obs_dim, seed, **kw):
    return PPO(net=DiscreteActorCritic(obs_dim, 5, 64, device=DEV, seed=seed), seed=seed, **kw)


def test_joint_rows_forward_independent_policies():
    N, D, R = 3, 18, 50
    env = Env(N)
    algos = [_ppo(D, 10 + i) for i in range(N)]
    pol = MultiAgentPolicy({a: algos[i] for i, a in enumerate(env.agents)}, env.agent_idx)
    assert pol.shared_policy is None
    obs = torch.randn(R, N, D, device=DEV)
    for a in algos:
        a.deterministic_eval = True  # greedy: comparable across calls
    out = pol(Batch(obs=obs))
    assert out.act.shape == (R, N) and out.policy.logp.shape == (R, N)
    for i, a in enumerate(algos):
        ref = ops.policy_forward(a.net.flat.data, obs[:, i].contiguous(), 5, 64, mode="mode")
        assert torch.equal(out.act[:, i].to(torch.int32), ref["act"])
        assert torch.equal(out.policy.v_s[:, i], ref["value"])
    # parallel-mode observation container (EnhancedPettingZooEnv layout) gives the same rows
    cont = Batch(observations=Batch({a: obs[:, i].cpu().numpy() for i, a in enumerate(env.agents)}),
                 agent_ids=np.array([env.agents] * R, dtype=object))
    out2 = pol(Batch(obs=cont))
    assert torch.equal(out2.act, out.act)


def test_collector_with_shared_manager_equals_plain_ppo():
    n_env, N, T = 32, 3, 25
    stores = []
    for wrap in (False, True):
        env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=5)
        algo = _ppo(env.obs_dim, 5)
        buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=DEV)
        policy = FlexibleMultiAgentPolicyManager(algo, env, mode="shared") if wrap else algo
        col = Collector(policy, env, buf, fused_rollout=False, use_graph=False)
        col.reset()
        with policy_within_training_step(policy):
            st = col.collect(n_step=n_env * T)
        assert st.n_collected_episodes == n_env
        stores.append((buf.obs_store.clone(), buf.act_store.clone(), buf.rew_store.clone(), buf.logp_store.clone()))
    for x, y in zip(*stores):
        assert torch.equal(x, y)


def test_independent_algorithms_update_only_on_their_agents_rows():
    n_env, N, T = 16, 3, 25
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=7)

    def make():
        return [_ppo(env.obs_dim, 20 + i, use_graph=False, shuffle="numpy") for i in range(N)]

    algos = make()
    ma = MultiAgentOnPolicyAlgorithm(algorithms=algos, env=env)
    assert ma.get_algorithm("agent_1") is algos[1] and ma.policy.shared_policy is None
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=DEV)
    col = Collector(ma, env, buf, use_graph=False)
    col.reset()
    with pytest.raises(RuntimeError, match="outside of a training step"):
        ma.update(buf, 128, 1)
    with policy_within_training_step(ma):
        col.collect(n_step=n_env * T)
        before = [a.net.flat.data.clone() for a in algos]
        np.random.seed(11)
        stats = ma.update(buf, batch_size=128, repeat=2)
    d = stats.get_loss_stats_dict()
    assert {f"agent_{i}/loss" for i in range(N)} <= set(d) and all(np.isfinite(v) for v in d.values())
    n_rows, per = n_env * T, 128
    assert d["agent_0/gradient_steps"] == 2 * (n_rows // per)  # 400 rows -> 3 minibatches (merge_last) x 2 repeats
    assert all(not torch.equal(b, a.net.flat.data) for b, a in zip(before, algos))
    # agent 1's algorithm, run alone on agent 1's lanes with the same permutation stream position, gives the same weights
    ref = make()
    for r, b in zip(ref, before):
        r.net.flat.data.copy_(b)
        r.net.sync_image()
    np.random.seed(11)
    for i in range(N):  # same order as the dispatcher: agent_0, agent_1, agent_2
        with policy_within_training_step(ref[i]):
            pb = ref[i]._preprocess_batch(buf)
            ref[i]._update_with_batch(pb, 128, 2, agent=i, buffer=buf)
    for r, a in zip(ref, algos):
        assert torch.equal(r.net.flat.data, a.net.flat.data)
    # checkpoint round trip keyed by agent id
    sd = ma.state_dict()
    assert set(sd) == {"agent_0", "agent_1", "agent_2"}
    algos[0].net.flat.data.zero_()
    ma.load_state_dict(sd)
    assert torch.equal(algos[0].net.flat.data, ref[0].net.flat.data)


def test_shared_algorithm_through_dispatcher_uses_the_fused_update():
    n_env, N, T = 16, 3, 25
    outs = []
    for wrap in (False, True):
        env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=9)
        algo = _ppo(env.obs_dim, 9, shuffle="numpy")
        top = MultiAgentOnPolicyAlgorithm(algorithms=[algo] * N, env=env) if wrap else algo
        buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=DEV)
        col = Collector(top, env, buf)
        col.reset()
        with policy_within_training_step(top):
            col.collect(n_step=n_env * T)
            np.random.seed(3)
            top.update(buf, batch_size=200, repeat=1)
        outs.append(algo.net.flat.data.clone())
    assert torch.equal(outs[0], outs[1])


def test_trainers_learn_from_device_agent_batches():
    n_env, N, T = 8, 3, 25
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=2)
    algos = {a: _ppo(env.obs_dim, 30 + i, use_graph=False) for i, a in enumerate(env.agents)}
    mgr = FlexibleMultiAgentPolicyManager(algos, env, mode="independent")
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=DEV)
    col = Collector(mgr, env, buf, use_graph=False)
    col.reset()
    with policy_within_training_step(mgr):
        col.collect(n_step=n_env * T)
    batch = agent_batches_from_buffer(buf, env.agents)
    idx = buf.sample_indices(0)
    host = buf[idx]
    assert batch.agent_1.obs.is_cuda and batch.agent_1.obs.shape == (n_env * T, env.obs_dim)
    assert np.array_equal(batch.agent_1.obs.cpu().numpy(), host.obs[:, 1])
    assert np.array_equal(batch.agent_2.rew.cpu().numpy(), host.rew[:, 2].astype(np.float32))
    assert np.array_equal(batch.global_obs.cpu().numpy(), host.obs.reshape(n_env * T, -1))  # "concatenate" global state
    before = {a: p.net.flat.data.clone() for a, p in algos.items()}
    tr = SimultaneousTrainer(mgr, agent_train_freq={"agent_2": 2})
    losses = tr.train_step(batch)
    assert set(losses) == {"agent_0", "agent_1"} and all(np.isfinite(v["loss"]) for v in losses.values())
    assert not torch.equal(before["agent_0"], algos["agent_0"].net.flat.data)
    assert torch.equal(before["agent_2"], algos["agent_2"].net.flat.data)  # trains every 2nd step only
    tr.train_step(batch)
    assert not torch.equal(before["agent_2"], algos["agent_2"].net.flat.data)


# ---- BASELINE configs[0]: 1-env DummyVectorEnv of a parallel-mode env through the reference-style host loop -------
class HostSpread:
    """ParallelEnv-shaped simple_spread on the host (the numpy oracle world; pettingzoo itself is not installed)."""

    metadata = {"name": "simple_spread_host"}

    def __init__(self, n_agent=3, max_cycles=25, seed=0):
        import sys

        sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
        import mpe_oracle

        self.world = mpe_oracle.SimpleSpreadWorld(n_agent, max_cycles=max_cycles, seed=seed)
        self.possible_agents = [f"agent_{i}" for i in range(n_agent)]
        self.observation_spaces = {a: Box(-np.inf, np.inf, (6 * n_agent,)) for a in self.possible_agents}
        self.action_spaces = {a: Discrete(5) for a in self.possible_agents}

    def _d(self, arr):
        return {a: arr[i] for i, a in enumerate(self.possible_agents)}

    def reset(self, seed=None, **kw):
        obs = self.world.reset()
        return self._d(obs.astype(np.float32)), {a: {} for a in self.possible_agents}

    def step(self, actions):
        act = np.array([actions[a] for a in self.possible_agents])
        obs, rew, term, trunc = self.world.step(act)
        return (self._d(obs.astype(np.float32)), self._d(rew), self._d(np.asarray(term, bool)),
                self._d(np.asarray(trunc, bool)), {a: {} for a in self.possible_agents})

    def close(self):
        pass


@pytest.mark.parametrize("n_env", [1, 3])
def test_host_collect_loop_with_dummy_vector_env(n_env):
    N, T = 3, 25
    venv = DummyVectorEnv([lambda i=i: EnhancedPettingZooEnv(HostSpread(N, T, seed=i), mode="parallel") for i in range(n_env)])
    algo = _ppo(6 * N, 4)
    buf = DeviceVectorReplayBuffer(n_env * 2 * T, n_env, N, 6 * N, device=DEV)
    seen = []
    col = Collector(algo, venv, buf, on_step_hook=lambda b: seen.append(len(b.rew)))
    assert not col._device_path
    col.reset()
    with policy_within_training_step(algo):
        st = col.collect(n_step=n_env * T)
    assert st.n_collected_steps == n_env * T and st.n_collected_episodes == n_env
    assert st.lens.tolist() == [T] * n_env and len(seen) == T and len(buf) == n_env * T
    # episode return bookkeeping (HIP) == sum of the stored per-agent rewards
    host = buf[buf.sample_indices(0)]
    per_env = host.rew.reshape(n_env, T, N).sum(1)
    np.testing.assert_allclose(st.returns.reshape(n_env, N), per_env, rtol=1e-6)
    assert host.truncated.reshape(n_env, T, N)[:, -1].all() and not host.truncated.reshape(n_env, T, N)[:, :-1].any()
    # obs_next of step t is obs of step t+1 inside an episode
    o, on = host.obs.reshape(n_env, T, N, -1), host.obs_next.reshape(n_env, T, N, -1)
    assert np.array_equal(on[:, :-1], o[:, 1:])
    # stored policy outputs are the ones the policy produced for these rows
    ref = ops.policy_forward(algo.net.flat.data, torch.as_tensor(host.obs).to(DEV).reshape(-1, 6 * N), 5, 64, mode="given",
                             act=torch.as_tensor(host.act).to(DEV, torch.int32).reshape(-1))
    np.testing.assert_allclose(host.policy.logp.reshape(-1), ref["logp"].cpu().numpy(), rtol=1e-6, atol=1e-7)
    # random collection and n_episode collection run through the same loop
    with policy_within_training_step(algo):
        st = col.collect(n_episode=n_env, random=True, reset_before_collect=True)
    assert st.n_collected_episodes == n_env
    with policy_within_training_step(algo):
        stats = algo.update(buf, batch_size=64, repeat=1)
    assert np.isfinite(list(stats.get_loss_stats_dict().values())).all()


# ---- the reference's own Collector known-answer sequence (test/base/test_collector.py:151-230) -----------------------
class MoveToRight:
    """Index walks right by action 1; reaching `size` terminates with reward 1 (reference test/base/env.py)."""

    def __init__(self, size):
        self.size, self.index = size, 0
        self.action_space = Discrete(2)
        self.observation_space = Box(0, size - 1, (1,))

    def reset(self, seed=None, **kw):
        self.index = 0
        return np.array([self.index], np.float32), {"key": 1}

    def step(self, action):
        self.index = self.index + 1 if int(action) == 1 else max(0, self.index - 1)
        done = self.index == self.size
        return np.array([self.index], np.float32), int(done), done, False, {"key": 1}

    def close(self):
        pass


class MaxActionPolicy(torch.nn.Module if torch.cuda.is_available() else object):
    def forward(self, batch, state=None, **kw):
        return Batch(act=np.ones(len(batch.obs)), state=state)


def test_collector_known_answer_layout_of_the_reference():
    venv = DummyVectorEnv([lambda s=s: MoveToRight(s) for s in (2, 3, 4, 5)])
    buf = DeviceVectorReplayBuffer(100, 4, n_agent=1, obs_dim=1, device=DEV)
    col = Collector(MaxActionPolicy(), venv, buf)
    col.reset()
    st = col.collect(n_step=8)
    assert st.n_collected_steps == 8 and st.n_collected_episodes == 1  # env 0 (size 2) finished once
    obs = np.zeros(100)
    obs[[0, 1, 25, 26, 50, 51, 75, 76]] = [0, 1, 0, 1, 0, 1, 0, 1]
    assert np.allclose(buf.obs[:, 0, 0], obs)
    assert np.allclose(buf[:].obs_next[..., 0, 0], [1, 2, 1, 2, 1, 2, 1, 2])
    rews = np.zeros(100)
    rews[[0, 1, 25, 26, 50, 51, 75, 76]] = [0, 1, 0, 0, 0, 0, 0, 0]
    assert np.allclose(buf.rew[:, 0], rews)
    # 4 more episodes: env 0 was reset (2 steps), the others finish what they started (1, 2, 3 steps): 8 + 8 rows
    st = col.collect(n_episode=4)
    assert st.n_collected_episodes == 4 and len(buf) == 16
    assert sorted(st.lens.tolist()) == [2, 3, 4, 5] and np.allclose(st.returns, 1.0)
    obs[[2, 3, 27, 52, 53, 77, 78, 79]] = [0, 1, 2, 2, 3, 2, 3, 4]
    assert np.allclose(buf.obs[:, 0, 0], obs)
    assert np.allclose(buf[:].obs_next[..., 0, 0], [1, 2, 1, 2, 1, 2, 3, 1, 2, 3, 4, 1, 2, 3, 4, 5])
    rews[[2, 3, 27, 52, 53, 77, 78, 79]] = [0, 1, 1, 0, 1, 0, 0, 1]
    assert np.allclose(buf.rew[:, 0], rews)
    assert np.array_equal(buf.sample_indices(0), [0, 1, 2, 3, 25, 26, 27, 50, 51, 52, 53, 75, 76, 77, 78, 79])
    col.collect(n_episode=4, random=True)
    # fresh start, 8 episodes: the short envs finish more of them (3 + 2 + 2 + 1), 25 rows in all -- the reference's
    # final obs layout [0..5], [25..30], [50..57], [75..79] (test_collector.py:214-218)
    col.reset_env()
    col.reset_buffer()
    assert col.collect(n_episode=8).n_collected_episodes == 8
    live = buf[:]
    per_env = {0: [0, 1, 0, 1, 0, 1], 1: [0, 1, 2, 0, 1, 2], 2: [0, 1, 2, 3, 0, 1, 2, 3], 3: [0, 1, 2, 3, 4]}
    assert np.allclose(live.obs[..., 0, 0], sum(per_env.values(), []))
    assert np.array_equal(buf.sample_indices(0), [*range(0, 6), *range(25, 31), *range(50, 58), *range(75, 80)])
    assert np.allclose(live.rew[:, 0], [0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 1])
    with pytest.raises(ValueError):
        Collector(MaxActionPolicy(), venv, DeviceVectorReplayBuffer(9, 3, n_agent=1, obs_dim=1, device=DEV))


def test_collector_lens_count_the_rows_in_the_buffer_as_the_reference_does(golden_dir):
    """`CollectStats.lens` is `len(episode_batch)` in the reference (collector.py:203,990-993): the episode's rows IN THE BUFFER.
    Behind `reset_buffer(keep_statistics=True)` -- what the trainer calls after every update (trainer.py:1104) -- an episode that
    was running counts its rows since the reset while its return stays whole.  The REFERENCE's own run (collector_port.npz:
    three collect(n_step) calls on the truncating MoveToRight envs with that reset in between) replayed by the host path of
    `Collector`: statistics, counters and every buffer row after each call.  (The CPU baseline port is pinned to the same
    fixture: tests/test_oracle_golden.py.)"""
    import sys

    sys.path.insert(0, golden_dir)
    from collector_script import env_step, scripted_action

    g = np.load(os.path.join(golden_dir, "collector_port.npz"))

    class Env:
        def __init__(self, size, limit):
            self.size, self.limit, self.index, self.steps = size, limit, 0, 0
            self.action_space, self.observation_space = Discrete(2), Box(0, size, (1,))

        def reset(self, seed=None, **kw):
            self.index, self.steps = 0, 0
            return np.array([self.index], np.float32), {}

        def step(self, action):
            self.index, self.steps, rew, term, trunc = env_step(self.index, self.steps, self.size, self.limit, action)
            return np.array([self.index], np.float32), rew, term, trunc, {}

        def close(self):
            pass

    class ScriptPolicy(torch.nn.Module):
        calls = 0

        def forward(self, batch, state=None, **kw):
            self.calls += 1
            return Batch(act=scripted_action(self.calls, np.asarray(batch.obs)))

    n_env = len(g["sizes"])
    venv = DummyVectorEnv([lambda s=int(s), l=int(l): Env(s, l) for s, l in zip(g["sizes"], g["limits"])])
    buf = DeviceVectorReplayBuffer(n_env * 8, n_env, n_agent=1, obs_dim=1, device=DEV)
    pol = ScriptPolicy()
    col = Collector(pol, venv, buf)
    col.reset()
    straddled = 0
    for i, n in enumerate(g["n_steps"]):
        if i:
            col.reset_buffer(keep_statistics=True)
        st = col.collect(n_step=int(n))
        assert (st.n_collected_steps, st.n_collected_episodes) == (int(g[f"c{i}_steps"]), int(g[f"c{i}_episodes"])), i
        assert np.array_equal(st.lens, g[f"c{i}_lens"]) and np.array_equal(st.returns, g[f"c{i}_returns"]), i
        assert [col.collect_step, col.collect_episode, pol.calls] == g[f"c{i}_counters"].tolist(), i
        idx = buf.sample_indices(0)
        assert np.array_equal(idx, g[f"c{i}_indices"]), i
        b = buf[idx]
        assert np.array_equal(b.obs[:, 0, 0], g[f"c{i}_obs"][:, 0]) and np.array_equal(b.obs_next[:, 0, 0], g[f"c{i}_obs_next"][:, 0]), i
        assert np.array_equal(b.act[:, 0], g[f"c{i}_act"]) and np.array_equal(b.rew[:, 0], g[f"c{i}_rew"]), i
        assert np.array_equal(b.terminated[:, 0], g[f"c{i}_terminated"]) and np.array_equal(b.truncated[:, 0], g[f"c{i}_truncated"]), i
        straddled += int((g[f"c{i}_lens"] == 1).sum())
    assert straddled >= 2   # (episodes of one buffered row exist only because of the reset: MoveToRight needs >= 2 steps)


def test_collector_replays_the_reference_run(golden_dir):
    """The synchronous `Collector` (collector.py:770-1098) against the REFERENCE's own run (tests/golden/collector.npz, made by
    make_fixtures.py::make_collector): five MoveToRight envs, two of them truncated by a step limit, a policy whose actions and
    `policy` tags follow one script on both sides (tests/golden/collector_script.py).  Ten calls -- n_step (a multiple of the env
    count and not), n_episode with fewer and with more episodes than envs (surplus-env removal, the env reset after an n_episode
    call), reset_before_collect, reset_buffer, reset_stat -- and after each: collected steps / episodes, episode lengths and
    returns in the reference's order, their statistics, the buffer length, the lifetime counters, the number of policy calls and
    the observations the next call starts from; at the end every buffer row.  Bit-exact."""
    import sys

    sys.path.insert(0, golden_dir)
    from collector_script import PLAN, env_step, scripted_action

    g = np.load(os.path.join(golden_dir, "collector.npz"))

    class Env:
        def __init__(self, size, limit):
            self.size, self.limit, self.index, self.steps = size, limit, 0, 0
            self.action_space, self.observation_space = Discrete(2), Box(0, size, (1,))

        def reset(self, seed=None, **kw):
            self.index, self.steps = 0, 0
            return np.array([self.index], np.float32), {"key": 1}

        def step(self, action):
            self.index, self.steps, rew, term, trunc = env_step(self.index, self.steps, self.size, self.limit, action)
            return np.array([self.index], np.float32), rew, term, trunc, {"key": 1}

        def close(self):
            pass

    class ScriptPolicy(torch.nn.Module):
        calls = 0

        def forward(self, batch, state=None, **kw):
            self.calls += 1
            obs = np.asarray(batch.obs)
            n = len(obs)
            return Batch(act=scripted_action(self.calls, obs),
                         policy=Batch(logp=(self.calls * 10.0 + np.arange(n)).astype(np.float32).reshape(n, 1)))

    n_env = len(g["sizes"])
    venv = DummyVectorEnv([lambda s=int(s), l=int(l): Env(s, l) for s, l in zip(g["sizes"], g["limits"])])
    buf = DeviceVectorReplayBuffer(400, n_env, n_agent=1, obs_dim=1, device=DEV)
    pol = ScriptPolicy()
    col = Collector(pol, venv, buf)
    col.reset()
    for i, (kind, n, extra) in enumerate(PLAN):
        kw = {}
        if extra == "reset_before_collect":
            kw["reset_before_collect"] = True
        elif extra == "reset_buffer":
            col.reset_buffer()
        elif extra == "reset_stat":
            col.reset_stat()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (n_step not a multiple of the env count, fewer episodes than envs: as in the reference run)
            st = col.collect(**{kind: n}, **kw)
        assert (st.n_collected_steps, st.n_collected_episodes) == (int(g[f"c{i}_steps"]), int(g[f"c{i}_episodes"])), i
        assert np.array_equal(st.lens, g[f"c{i}_lens"]) and np.array_equal(st.returns, g[f"c{i}_returns"]), i
        assert len(buf) == int(g[f"c{i}_len_buf"]), i
        assert [col.collect_step, col.collect_episode, pol.calls] == g[f"c{i}_counters"].tolist(), i
        assert np.array_equal(np.asarray(col._pre_obs, np.float32).reshape(-1), g[f"c{i}_pre_obs"]), i
        if f"c{i}_ret_stat" in g:
            rs, ls = st.returns_stat, st.lens_stat
            assert np.allclose([rs.mean, rs.std, rs.max, rs.min], g[f"c{i}_ret_stat"], rtol=1e-12, atol=0), i
            assert np.allclose([ls.mean, ls.std, ls.max, ls.min], g[f"c{i}_len_stat"], rtol=1e-12, atol=0), i
        else:
            assert st.returns_stat is None and st.lens_stat is None, i
    idx = buf.sample_indices(0)
    assert np.array_equal(idx, g["indices"])
    b = buf[idx]
    assert np.array_equal(b.obs[:, 0, 0], g["obs"][:, 0]) and np.array_equal(b.obs_next[:, 0, 0], g["obs_next"][:, 0])
    assert np.array_equal(b.act[:, 0], g["act"]) and np.array_equal(b.rew[:, 0], g["rew"])
    assert np.array_equal(b.terminated[:, 0], g["terminated"]) and np.array_equal(b.truncated[:, 0], g["truncated"])
    assert np.array_equal(b.done, g["done"]) and g["truncated"].sum() >= 4 and g["terminated"].sum() >= 4
    assert np.array_equal(b.policy.logp[:, 0], g["policy_tag"])
    assert np.array_equal(np.asarray(buf.last_index), g["last_index"])


def test_async_collector_replays_the_reference_run(golden_dir):
    """`AsyncCollector` (collector.py:1116-1394) against the REFERENCE's own run (tests/golden/async_collector.npz): four envs of
    lengths 2..5 behind an async vector env (wait_num 3) whose readiness follows one script on both sides
    (tests/golden/async_script.py), a policy that tags every action with (forward call, position).  After each of eight
    collect(n_episode / n_step) calls: collected steps and episodes, episode lengths and returns, the ready set and the envs
    still stepping; at the end every buffer row -- obs, act, the policy tag handed out for THAT env, obs_next, reward, flags --
    in the reference's `sample_indices(0)` order, and the collector's counters.  Bit-exact (integers and small floats)."""
    import sys

    sys.path.insert(0, golden_dir)
    from async_script import scripted_ready

    from tianshou_marl_amd.data import AsyncCollector

    g = np.load(os.path.join(golden_dir, "async_collector.npz"))

    class Env(MoveToRight):
        def step(self, action):
            obs, _, done, trunc, info = super().step(action)
            return [obs, float(done) * (self.size + 1), done, trunc, info]  # (the fixture's reward: size + 1 at the end)

    class TagPolicy(torch.nn.Module):
        calls = 0

        def forward(self, batch, state=None, **kw):
            self.calls += 1
            n = len(batch.obs)
            return Batch(act=np.ones(n, np.int64), policy=Batch(logp=(self.calls * 10.0 + np.arange(n)).astype(np.float32).reshape(n, 1)))

    venv = DummyVectorEnv([lambda s=int(s): Env(s) for s in g["sizes"]], wait_num=int(g["wait_num"]))
    assert venv.is_async
    calls = [0]

    def selector(waiting, wait_num):
        pos = scripted_ready(len(waiting), wait_num, calls[0])
        calls[0] += 1
        return pos

    venv.ready_selector = selector
    sent, returned, orig_step = [], [], venv.step

    def logged_step(action, id=None):  # noqa: A002  (the interleaving itself: env ids handed to / returned by every step call)
        out = orig_step(action, id)
        sent.extend([*(np.asarray(id).tolist() if id is not None else []), -1])
        returned.extend([*(int(i["env_id"]) for i in out[-1]), -1])
        return out

    venv.step = logged_step
    buf = DeviceVectorReplayBuffer(240, 4, n_agent=1, obs_dim=1, device=DEV)
    with pytest.warns(UserWarning, match="extra transitions"):
        col = AsyncCollector(TagPolicy(), venv, buf)
    col.reset()
    with pytest.raises(ValueError):
        col.collect()
    for i, (kind, n) in enumerate(zip(g["plan_kind"], g["plan_n"])):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (n_step not a multiple of the env count: as in the reference run)
            st = col.collect(**{str(kind): int(n)})
        assert (st.n_collected_steps, st.n_collected_episodes) == (int(g[f"c{i}_steps"]), int(g[f"c{i}_episodes"])), i
        assert np.array_equal(st.lens, g[f"c{i}_lens"]) and np.array_equal(st.returns, g[f"c{i}_returns"]), i
        assert np.array_equal(col._ready, g[f"c{i}_ready"]) and sorted(venv.waiting_id) == g[f"c{i}_waiting"].tolist(), i
        assert len(buf) == int(g[f"c{i}_len_buf"]), i
    assert calls[0] == int(g["wait_calls"])
    assert sent == g["trace_sent"].tolist() and returned == g["trace_returned"].tolist()
    assert (col.collect_step, col.collect_episode) == (int(g["collect_step"]), int(g["collect_episode"]))
    idx = buf.sample_indices(0)
    assert np.array_equal(idx, g["indices"])
    b = buf[idx]
    assert np.array_equal(b.obs[:, 0, 0], g["obs"][:, 0]) and np.array_equal(b.obs_next[:, 0, 0], g["obs_next"][:, 0])
    assert np.array_equal(b.act[:, 0], g["act"]) and np.array_equal(b.rew[:, 0], g["rew"])
    assert np.array_equal(b.terminated[:, 0], g["terminated"]) and np.array_equal(b.truncated[:, 0], g["truncated"])
    assert np.array_equal(b.done, g["done"])
    assert np.array_equal(b.policy.logp[:, 0], g["policy_tag"])          # the entry handed out for that env at action time
    assert np.array_equal(idx // (240 // 4), g["env_id"])                # rows sit in their env's sub-buffer
    # a fresh reset fetches the env that is still stepping before it resets (collector.py:1188-1196)
    assert venv.waiting_id
    col.reset()
    assert venv.waiting_id == [] and np.array_equal(col._ready, np.arange(4)) and len(buf) == 0


@pytest.mark.parametrize("shuffle,batch_size,repeat,opts", [
    ("device", None, 1, {}), ("device", 100, 2, dict(max_grad_norm=0.5, value_clip=True)), ("numpy", 64, 1, dict(dual_clip=2.0)),
    ("device", 256, 3, dict(advantage_normalization=False))])
def test_learn_as_one_graph_replay_equals_eager_launches(shuffle, batch_size, repeat, opts):
    """PPO.learn(batch) -- what the MARL trainers call per policy and step (training_coordinator.py:118,154,336) -- replays
    ONE hipGraph per call; parameters, optimizer state and the returned statistics must equal the eager launch sequence
    bit for bit over several calls (device-resident step count, permutation counter and learning rate advance on replay)."""
    from tianshou_marl_amd.algorithm.optim import LambdaLR

    n, D = 600, 18
    outs = []
    for use_graph in (True, False):
        algo = PPO(net=DiscreteActorCritic(D, 5, 64, device=DEV, seed=3), seed=9, lr=1e-3, shuffle=shuffle, use_graph=use_graph,
                   **opts)
        algo.lr_schedulers.append(LambdaLR(algo, lambda e: 1.0 - 0.2 * e))  # the learning rate moves between calls
        np.random.seed(4)
        g = torch.Generator().manual_seed(1)
        res = []
        for it in range(3):
            b = Batch(obs=torch.randn(n, D, generator=g).numpy(), act=torch.randint(0, 5, (n,), generator=g).numpy(),
                      rew=torch.randn(n, generator=g).numpy(), obs_next=torch.randn(n, D, generator=g).numpy(),
                      terminated=(torch.rand(n, generator=g) < 0.05).numpy(), truncated=(torch.rand(n, generator=g) < 0.05).numpy())
            res.append(algo.learn(b, batch_size=batch_size, repeat=repeat))
            for sch in algo.lr_schedulers:
                sch.step()
        outs.append((algo.net.flat.data.clone(), algo.exp_avg.clone(), algo.exp_avg_sq.clone(), algo.opt_step,
                     int(algo._perm_ctr.item()), res))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[3] == b[3] and a[4] == b[4] and a[5] == b[5]


def test_self_play_checkpoint_restores_the_opponent_pool(tmp_path):
    """SelfPlayTrainer.save_checkpoint / load_checkpoint (training_coordinator.py:225-263, 545-571) with the opponent pool
    the reference leaves out: after a reload into fresh objects the pool holds the same frozen snapshots (parameters and
    optimizer state), in order, with their win rates, and the learner continues bit-identically."""
    from tianshou_marl_amd.algorithm.multiagent import SelfPlayTrainer

    def build():
        env = type("E", (), {"agents": ["a", "b"]})()
        pols = {"a": PPO(net=DiscreteActorCritic(6, 5, 64, device=DEV, seed=1), seed=1, shuffle="device"),
                "b": PPO(net=DiscreteActorCritic(6, 5, 64, device=DEV, seed=2), seed=2, shuffle="device")}
        mgr = FlexibleMultiAgentPolicyManager(pols, env, mode="independent")
        return mgr, SelfPlayTrainer(mgr, main_agent_id="a", snapshot_interval=1, opponent_pool_size=3)

    g = torch.Generator().manual_seed(0)
    mk = lambda n: Batch(obs=torch.randn(n, 6, generator=g).numpy(), act=torch.randint(0, 5, (n,), generator=g).numpy(),  # noqa: E731
                         rew=torch.randn(n, generator=g).numpy(), obs_next=torch.randn(n, 6, generator=g).numpy(),
                         terminated=np.zeros(n, bool))
    batches = [Batch(a=mk(200), b=mk(200)) for _ in range(5)]
    mgr, tr = build()
    for b in batches[:4]:
        tr.train_step(b)
    assert len(tr.opponent_pool) == 3  # 4 snapshots taken, the oldest one dropped
    tr.update_win_rate(id(tr.opponent_pool[1]), True)
    path = str(tmp_path / "selfplay.pt")
    tr.save_checkpoint(path)
    mgr2, tr2 = build()
    tr2.load_checkpoint(path)
    assert tr2.step_count == tr.step_count and len(tr2.opponent_pool) == 3
    for p, q in zip(tr.opponent_pool, tr2.opponent_pool):
        assert torch.equal(p.net.flat.data, q.net.flat.data) and torch.equal(p.exp_avg_sq, q.exp_avg_sq) and not q.training
    assert tr2.opponent_win_rates[id(tr2.opponent_pool[1])] == pytest.approx(0.55)
    assert torch.equal(mgr.policies["a"].net.flat.data, mgr2.policies["a"].net.flat.data)
    l1, l2 = tr.train_step(batches[4]), tr2.train_step(batches[4])
    assert dict(l1["a"]) == dict(l2["a"])
    assert torch.equal(mgr.policies["a"].net.flat.data, mgr2.policies["a"].net.flat.data)
