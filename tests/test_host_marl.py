"""Host logic of the multi-agent layer (no GPU): policy-map construction and validation of
FlexibleMultiAgentPolicyManager, the training coordinators' scheduling, checkpoints, Elo / win-rate
bookkeeping, MapTrainingStats key prefixes.  Modelled on the reference's
test/multiagent/test_flexible_policy_manager.py and test_training_coordination.py (mock policies
that count `learn` calls)."""
import numpy as np
import pytest
import torch

from tianshou_marl_amd.algorithm.multiagent import (
    FlexibleMultiAgentPolicyManager,
    LeaguePlayTrainer,
    MapTrainingStats,
    MATrainer,
    SelfPlayTrainer,
    SequentialTrainer,
    SimultaneousTrainer,
)
from tianshou_marl_amd.data import Batch
from tianshou_marl_amd.data.stats import A2CTrainingStats, SequenceSummaryStats


class Env:
    def __init__(self, n):
        self.agents = [f"agent_{i}" for i in range(n)]
        self.agent_idx = {a: i for i, a in enumerate(self.agents)}


class CountingPolicy(torch.nn.Module):
    def __init__(self, tag=0):
        super().__init__()
        self.w = torch.nn.Parameter(torch.full((2,), float(tag)))
        self.learn_calls = 0
        self.seen = []
        self.device = torch.device("cpu")

    def forward(self, batch, state=None, **kw):
        return Batch(act=np.zeros(len(batch.obs), int), state=None)

    def learn(self, batch, **kw):
        self.learn_calls += 1
        self.seen.append(batch)
        return {"loss": float(self.learn_calls)}


def agent_batch(n=4):
    return Batch(obs=np.zeros((n, 2), np.float32), act=np.zeros(n, int), rew=np.ones(n), obs_next=np.zeros((n, 2), np.float32),
                 terminated=np.zeros(n, bool))


def ma_batch(env, with_global=False):
    b = Batch({a: agent_batch() for a in env.agents})
    if with_global:
        b.global_obs = np.ones((4, 6), np.float32)
        b.global_obs_next = np.ones((4, 6), np.float32) * 2
    return b


# ---- FlexibleMultiAgentPolicyManager -----------------------------------------------------------------------
def test_manager_independent_list_and_dict():
    env = Env(3)
    pols = [CountingPolicy(i) for i in range(3)]
    m = FlexibleMultiAgentPolicyManager(pols, env, mode="independent")
    assert [m.policy_map[a] for a in env.agents] == pols
    assert len(m.policies) == 3 and m.policy_mapping is m.policy_map
    assert not m.get_shared_parameters()
    assert m.get_policy_groups() == {"group_0": ["agent_0"], "group_1": ["agent_1"], "group_2": ["agent_2"]}
    d = {a: CountingPolicy() for a in env.agents}
    m = FlexibleMultiAgentPolicyManager(d, env, mode="independent")
    assert all(m.policy_map[a] is d[a] for a in env.agents)


def test_manager_shared_grouped_custom():
    env = Env(4)
    p = CountingPolicy()
    m = FlexibleMultiAgentPolicyManager(p, env, mode="shared")
    assert all(m.policy_map[a] is p for a in env.agents)
    assert list(m.policies) == ["shared"] and m.get_shared_parameters()
    assert m.get_policy_groups() == {"shared": env.agents}
    assert m.shared_policy is p
    pols = {"red": CountingPolicy(1), "blue": CountingPolicy(2)}
    groups = {"red": ["agent_0", "agent_1"], "blue": ["agent_2", "agent_3"]}
    m = FlexibleMultiAgentPolicyManager(pols, env, mode="grouped", agent_groups=groups)
    assert m.policy_map["agent_1"] is pols["red"] and m.policy_map["agent_3"] is pols["blue"]
    assert set(m.policies) == {"red", "blue"} and m.shared_policy is None
    assert m.get_policy_groups() == {"group_0": ["agent_0", "agent_1"], "group_1": ["agent_2", "agent_3"]}
    roles = {"explorer": CountingPolicy(), "defender": CountingPolicy()}
    m = FlexibleMultiAgentPolicyManager(roles, env, mode="custom",
                                        policy_mapping_fn=lambda a: "explorer" if int(a[-1]) % 2 == 0 else "defender")
    assert m.policy_map["agent_2"] is roles["explorer"] and m.policy_map["agent_3"] is roles["defender"]
    assert len(m.policies) == 2


@pytest.mark.parametrize("kw,msg", [
    (dict(policies="single", mode="independent"), "Independent mode requires list or dict"),
    (dict(policies="single", mode="grouped"), "agent_groups"),
    (dict(policies={}, mode="custom"), "policy_mapping_fn"),
    (dict(policies=[0, 1], mode="independent"), "must match number of agents"),
    (dict(policies={"agent_0": 0}, mode="independent"), "Missing policies"),
    (dict(policies={"a": 0}, mode="grouped", agent_groups={"a": ["agent_0"]}), "not assigned"),
    (dict(policies={"a": 0}, mode="grouped", agent_groups={"a": ["agent_0"], "b": ["agent_1", "agent_2"]}), "No policy found"),
    (dict(policies={"x": 0}, mode="custom", policy_mapping_fn=lambda a: "y"), "not found for agent"),
])
def test_manager_validation_errors(kw, msg):
    env = Env(3)
    pol = kw.pop("policies")
    if pol == "single":
        pol = CountingPolicy()
    elif isinstance(pol, list):
        pol = [CountingPolicy() for _ in pol]
    elif isinstance(pol, dict):
        pol = {k: CountingPolicy() for k in pol}
    with pytest.raises(ValueError, match=msg):
        FlexibleMultiAgentPolicyManager(pol, env, **kw)


def test_manager_training_flag_reaches_every_policy():
    env = Env(3)
    pols = [CountingPolicy(i) for i in range(3)]
    m = FlexibleMultiAgentPolicyManager(pols, env, mode="independent")
    assert m.is_within_training_step is False
    m.is_within_training_step = True
    assert all(p.is_within_training_step for p in pols)
    m.train(False)  # nn.Module plumbing reaches the sub-policies (the collector toggles it)
    assert not any(p.training for p in pols)


# ---- trainers ----------------------------------------------------------------------------------------------
def test_matrainer_modes_and_round_robin():
    env = Env(3)
    pols = {a: CountingPolicy() for a in env.agents}
    m = FlexibleMultiAgentPolicyManager(pols, env, mode="independent")
    with pytest.raises(ValueError, match="Invalid training mode"):
        MATrainer(m, "nope")
    tr = MATrainer(m, "simultaneous")
    out = tr.train_step(ma_batch(env, with_global=True))
    assert set(out) == set(env.agents) and tr.step_count == 1
    assert all(p.learn_calls == 1 for p in pols.values())
    assert "global_obs" in pols["agent_0"].seen[0] and "global_obs_next" in pols["agent_0"].seen[0]
    tr.set_training_mode("sequential")
    order = [next(iter(tr.train_step(ma_batch(env)))) for _ in range(4)]
    assert order == ["agent_0", "agent_1", "agent_2", "agent_0"]
    for mode in ("self_play", "league"):
        tr.set_training_mode(mode)
        with pytest.raises(NotImplementedError):
            tr.train_step(ma_batch(env))
    with pytest.raises(ValueError):
        tr.set_training_mode("bogus")


def test_simultaneous_trainer_frequency_and_shared():
    env = Env(3)
    pols = {a: CountingPolicy() for a in env.agents}
    m = FlexibleMultiAgentPolicyManager(pols, env, mode="independent")
    tr = SimultaneousTrainer(m, agent_train_freq={"agent_1": 2, "agent_2": 3})
    for _ in range(6):
        tr.train_step(ma_batch(env))
    assert [pols[a].learn_calls for a in env.agents] == [6, 3, 2]
    shared = CountingPolicy()
    m = FlexibleMultiAgentPolicyManager(shared, env, mode="shared")
    out = SimultaneousTrainer(m).train_step(ma_batch(env))
    assert set(out) == set(env.agents) and shared.learn_calls == 3  # one learn per agent batch on the shared policy


def test_sequential_trainer_order_steps_and_modes():
    env = Env(3)
    pols = {a: CountingPolicy() for a in env.agents}
    m = FlexibleMultiAgentPolicyManager(pols, env, mode="independent")
    tr = SequentialTrainer(m, agent_order=["agent_2", "agent_0"], steps_per_agent=2)
    seen = [next(iter(tr.train_step(ma_batch(env)))) for _ in range(5)]
    assert seen == ["agent_2", "agent_2", "agent_0", "agent_0", "agent_2"]
    assert pols["agent_2"].training and not pols["agent_0"].training and not pols["agent_1"].training
    assert pols["agent_1"].learn_calls == 0


def test_self_play_snapshots_sampling_and_state():
    env = Env(2)
    pols = {a: CountingPolicy() for a in env.agents}
    m = FlexibleMultiAgentPolicyManager(pols, env, mode="independent")
    tr = SelfPlayTrainer(m, main_agent_id="agent_0", snapshot_interval=2, opponent_pool_size=2, opponent_sampling="latest")
    assert tr._sample_opponent() is None
    for _ in range(6):
        out = tr.train_step(ma_batch(env))
        assert list(out) == ["agent_0"]
    assert pols["agent_1"].learn_calls == 0 and pols["agent_0"].learn_calls == 6
    assert len(tr.opponent_pool) == 2  # 3 snapshots taken, pool keeps the 2 newest
    assert tr.opponent_pool[-1] is not pols["agent_0"] and not tr.opponent_pool[-1].training
    assert tr._sample_opponent() is tr.opponent_pool[-1]
    oid = id(tr.opponent_pool[0])
    tr.update_win_rate(oid, True)
    assert tr.opponent_win_rates[oid] == pytest.approx(0.55)
    tr.update_win_rate(oid, False)
    assert tr.opponent_win_rates[oid] == pytest.approx(0.495)
    tr.opponent_sampling = "prioritized"
    np.random.seed(0)
    assert tr._sample_opponent() in tr.opponent_pool
    tr.opponent_sampling = "uniform"
    assert tr._sample_opponent() in tr.opponent_pool
    st = tr.state_dict()
    assert st["opponent_pool_size"] == 2 and st["main_agent_id"] == "agent_0" and st["step_count"] == 6
    tr2 = SelfPlayTrainer(m, main_agent_id="x")
    tr2.load_state_dict(st)
    assert tr2.main_agent_id == "agent_0" and tr2.step_count == 6
    # the opponent pool travels with the state (the reference leaves it out): same snapshots, in order, as frozen copies,
    # with their win rates (the live dict is keyed by object identity: compared in pool order)
    assert len(tr2.opponent_pool) == 2 and all(not p.training for p in tr2.opponent_pool)
    for p, q in zip(tr.opponent_pool, tr2.opponent_pool):
        assert q is not p and torch.equal(p.w, q.w)
    rates = lambda t: [t.opponent_win_rates.get(id(p), 0.5) for p in t.opponent_pool]  # noqa: E731
    assert rates(tr2) == pytest.approx(rates(tr)) and rates(tr)[0] == pytest.approx(0.495)


def test_league_matchmaking_elo_and_promotion():
    env = Env(4)
    pols = {a: CountingPolicy() for a in env.agents}
    m = FlexibleMultiAgentPolicyManager(pols, env, mode="independent")
    tr = LeaguePlayTrainer(m, matchmaking="random", games_per_evaluation=2)
    np.random.seed(1)
    out = tr.train_step(ma_batch(env))
    assert len(out) == 2 and set(out) <= set(env.agents)
    tr.update_match_result("agent_0", "agent_1")
    assert tr.agent_performance["agent_0"] == pytest.approx(0.55) and tr.agent_performance["agent_1"] == pytest.approx(0.45)
    assert tr.elo_ratings["agent_0"] == pytest.approx(1016.0) and tr.elo_ratings["agent_1"] == pytest.approx(984.0)
    assert list(tr.match_history) == [("agent_0", "agent_1")]
    for _ in range(3):
        tr.update_match_result("agent_0", "agent_1")
    promoted, relegated = tr._update_league()
    assert promoted == ["agent_0"] and relegated == ["agent_1"]
    tr.matchmaking = "elo"
    ranked = sorted(tr.league, key=lambda a: tr.elo_ratings[a])
    for _ in range(5):
        a, b = tr._make_match()
        assert abs(ranked.index(a) - ranked.index(b)) == 1  # neighbours in rating order
    tr.matchmaking = "win_rate"
    assert len(tr._make_match()) == 2
    tr.league = ["agent_0"]
    assert tr._make_match() == ["agent_0"]


def test_trainers_replay_the_reference_trace():
    """The four coordinators against a trace of the REFERENCE's own (tests/golden/trainers.npz, make_fixtures.py::make_trainers):
    mock policies, numpy's global generator seeded alike, the schedules of tests/golden/trainer_script.py.  Who learns at every
    step and how often, the sequential order and train / eval flags, which snapshot self-play samples under its three rules
    (the draws must come call for call as the reference's), win rates, every league match, Elo and performance value (exact
    float equality), promotion / relegation lists."""
    import os
    import sys

    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gd)
    import trainer_script as ts

    g = np.load(os.path.join(gd, "trainers.npz"))

    class P(CountingPolicy):
        version = -1

    env = Env(3)
    pols = {a: P() for a in env.agents}
    tr = SimultaneousTrainer(FlexibleMultiAgentPolicyManager(pols, env, mode="independent"), agent_train_freq={"agent_1": 2, "agent_2": 3})
    rows = []
    for _ in range(ts.SIMULTANEOUS_STEPS):
        losses = tr.train_step(ma_batch(env))
        rows.append([int(a in losses) for a in env.agents] + [pols[a].learn_calls for a in env.agents])
    assert np.array_equal(rows, g["simultaneous"])
    pols = {a: P() for a in env.agents}
    tr = SequentialTrainer(FlexibleMultiAgentPolicyManager(pols, env, mode="independent"), agent_order=["agent_2", "agent_0", "agent_1"],
                           steps_per_agent=2)
    rows = []
    for _ in range(ts.SEQUENTIAL_STEPS):
        losses = tr.train_step(ma_batch(env))
        rows.append([env.agents.index(next(iter(losses)))] + [int(pols[a].training) for a in env.agents])
    assert np.array_equal(rows, g["sequential"])
    # self-play
    env = Env(2)
    pols = {a: P() for a in env.agents}
    tr = SelfPlayTrainer(FlexibleMultiAgentPolicyManager(pols, env, mode="independent"), main_agent_id="agent_0",
                         snapshot_interval=ts.SNAPSHOT_INTERVAL, opponent_pool_size=ts.POOL_SIZE)
    np.random.seed(ts.SEED)
    for i in range(ts.SELFPLAY_STEPS):
        pols["agent_0"].version = i
        tr.opponent_sampling = ts.selfplay_sampling(i)
        losses = tr.train_step(ma_batch(env))
        opp = tr._sample_opponent()
        if opp is not None:
            tr.update_win_rate(id(opp), ts.selfplay_won(i))
        pool = [p.version for p in tr.opponent_pool] + [-1] * (ts.POOL_SIZE - len(tr.opponent_pool))
        row = [int("agent_0" in losses), pols["agent_0"].learn_calls, pols["agent_1"].learn_calls,
               -1 if opp is None else opp.version, len(tr.opponent_win_rates), *pool]
        assert row == g["selfplay"][i].tolist(), (i, row, g["selfplay"][i].tolist())
        rates = [tr.opponent_win_rates.get(id(p), -1.0) for p in tr.opponent_pool] + [-1.0] * (ts.POOL_SIZE - len(tr.opponent_pool))
        assert rates == g["selfplay_rates"][i].tolist(), i
    # league
    env = Env(ts.LEAGUE_AGENTS)
    pols = {a: P() for a in env.agents}
    tr = LeaguePlayTrainer(FlexibleMultiAgentPolicyManager(pols, env, mode="independent"), games_per_evaluation=ts.GAMES_PER_EVALUATION)
    np.random.seed(ts.SEED)
    for i in range(ts.LEAGUE_STEPS):
        tr.matchmaking = ts.league_matchmaking(i)
        before = [pols[a].learn_calls for a in env.agents]
        losses = tr.train_step(ma_batch(env))
        match = [a for a in env.agents if pols[a].learn_calls > before[env.agents.index(a)]]
        order = list(losses)
        w, l = (order[0], order[1]) if ts.league_winner_first(i) else (order[1], order[0])
        tr.update_match_result(w, l)
        row = [env.agents.index(order[0]), env.agents.index(order[1]), len(match), tr.game_count, len(tr.match_history)]
        assert row == g["league"][i].tolist(), (i, row, g["league"][i].tolist())
        assert [tr.elo_ratings[a] for a in env.agents] == g["league_elo"][i].tolist(), i
        assert [tr.agent_performance[a] for a in env.agents] == g["league_perf"][i].tolist(), i
        pr, rl = tr._update_league()
        assert [int(a in pr) for a in env.agents] + [int(a in rl) for a in env.agents] == g["league_lists"][i].tolist(), i


def test_trainer_checkpoint_roundtrip(tmp_path):
    env = Env(2)
    pols = {a: CountingPolicy(i + 1) for i, a in enumerate(env.agents)}
    m = FlexibleMultiAgentPolicyManager(pols, env, mode="independent")
    tr = MATrainer(m, "sequential")
    tr.train_step(ma_batch(env))
    path = str(tmp_path / "ckpt.pt")
    tr.save_checkpoint(path)
    with torch.no_grad():
        pols["agent_0"].w.zero_()
    tr2 = MATrainer(m, "simultaneous")
    tr2.load_checkpoint(path)
    assert tr2.step_count == 1 and tr2.training_mode == "sequential"
    assert pols["agent_0"].w.tolist() == [1.0, 1.0]


def test_map_training_stats_prefixes():
    def st(x):
        s = SequenceSummaryStats.from_sequence([x, x + 2])
        return A2CTrainingStats(loss=s, actor_loss=s, vf_loss=s, ent_loss=s, gradient_steps=2, train_time=x)

    ms = MapTrainingStats({"agent_0": st(1.0), "agent_1": st(3.0)})
    d = ms.get_loss_stats_dict()
    assert d["agent_0/loss"] == 2.0 and d["agent_1/vf_loss"] == 4.0 and d["agent_1/gradient_steps"] == 2.0
    assert ms.train_time == 3.0
    assert MapTrainingStats({"a": st(1.0), "b": st(3.0)}, train_time_aggregator="mean").train_time == 2.0
