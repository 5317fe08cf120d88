"""GPU tests (`-m gpu`) of the AEC-row path: the rows `PettingZooEnv` emits -- one agent's
turn per row, `obs = {agent_id, obs, mask}`, per-agent reward vector -- stored in the device buffer
(`DeviceAECReplayBuffer`), handed back in the reference's Batch layout, and trained on through the MARL dispatcher with
the reference's flat per-agent semantics (quirks Q1 / Q2 kept; pinned by tests/golden/marl_dispatch.npz, which the
reference itself produced)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from test_host_env import TinyAEC

    from tianshou_marl_amd.algorithm.multiagent import MARLDispatcher, MultiAgentOnPolicyAlgorithm
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data import Batch, Collector, DeviceAECReplayBuffer
    from tianshou_marl_amd.env.pettingzoo_env import PettingZooEnv
    from tianshou_marl_amd.env.venvs import DummyVectorEnv
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

DEV = "cuda"


def test_per_agent_returns_on_aec_rows_match_the_reference_incl_quirk_q1(golden_dir):
    """The fixture's buffer (2 envs, 2 agents taking turns, 3 rounds, nothing finished): per-agent
    `compute_episodic_return` with that agent's reward column, gamma = lambda = 1, zero values.  agent_0's returns in
    env 0 include env 1's (its last row is not the sub-buffer's last row: quirk Q1); agent_1's do not."""
    g = np.load(os.path.join(golden_dir, "marl_dispatch.npz"), allow_pickle=True)
    n_env, T = 2, 3
    agents = ["agent_0", "agent_1"]
    buf = DeviceAECReplayBuffer(n_env * T * 2, n_env, agents, obs_dim=2, device=DEV)
    k = 0
    for t in range(T):
        for a in range(2):
            k += 1
            rew = np.zeros((n_env, 2))
            rew[:, a] = np.arange(1, n_env + 1) * k
            ids = np.array([f"agent_{a}"] * n_env, dtype=object)
            buf.add(Batch(obs=Batch(agent_id=ids, obs=np.zeros((n_env, 2), np.float32)), act=np.zeros(n_env, int), rew=rew,
                          terminated=np.zeros(n_env, bool), truncated=np.zeros(n_env, bool),
                          obs_next=Batch(agent_id=ids, obs=np.zeros((n_env, 2), np.float32))), buffer_ids=np.arange(n_env))
    batch, indices = buf.sample(0)
    assert np.array_equal(indices, g["q1_all_indices"]) and np.array_equal(buf.unfinished_index(), g["q1_unfinished"])
    assert np.array_equal([buf.agent_idx[a] for a in batch.obs.agent_id], g["q1_agent_of_row"])
    assert batch.rew.shape == (12, 2) and batch.rew.dtype == np.float64 and batch.obs.obs.shape == (12, 2)
    idx_all, pos, offs = MARLDispatcher.aec_partition(buf)
    for a in range(2):
        rows = pos[int(offs[a]):int(offs[a + 1])]
        assert np.array_equal(rows.cpu().numpy(), g[f"q1_agent{a}_idx"])           # np.nonzero(agent_id == agent)
        tind = idx_all[rows].contiguous()
        assert np.array_equal(tind.cpu().numpy(), g[f"q1_agent{a}_indices"])
        z = torch.zeros(len(rows), device=DEV)
        ret, adv = MARLDispatcher.aec_returns(buf, tind, a, z, z, 1.0, 1.0)
        np.testing.assert_allclose(ret.cpu().numpy(), g[f"q1_agent{a}_returns"], rtol=1e-6)
    assert g["q1_agent0_returns"][2] != g["q1_agent1_returns"][2] - 1  # (the two agents really differ: Q1 bites agent_0)


def test_aec_pipeline_collects_stores_and_trains():
    """configs[0]-style plumbing on the reference's only runnable layout: DummyVectorEnv of PettingZooEnv (AEC) -> host
    Collector loop -> DeviceAECReplayBuffer -> MultiAgentOnPolicyAlgorithm.update through the dispatcher."""
    n_env, N, A, horizon = 4, 3, 4, 5
    venv = DummyVectorEnv([lambda: PettingZooEnv(TinyAEC(n=N, n_act=A, horizon=horizon)) for _ in range(n_env)])
    algos = [PPO(net=DiscreteActorCritic(2, A, 64, device=DEV, seed=10 + i), lr=1e-3, seed=i, shuffle="numpy", use_graph=False)
             for i in range(N)]
    env0 = PettingZooEnv(TinyAEC(n=N, n_act=A, horizon=horizon))  # (agents / agent_idx for the dispatcher, marl.py:197-203)
    ma = MultiAgentOnPolicyAlgorithm(algorithms=algos, env=env0)
    buf = DeviceAECReplayBuffer(n_env * 40, n_env, env0.agents, obs_dim=2, n_act=A, device=DEV)
    col = Collector(ma, venv, buf)
    col.reset()
    with policy_within_training_step(ma):
        st = col.collect(n_step=n_env * 32)  # 32 turns per env: two full episodes of 15 turns and a cut one
    assert st.n_collected_steps == n_env * 32 and st.n_collected_episodes == 2 * n_env
    assert st.returns.shape == (2 * n_env, N)  # per-agent episode returns (collector.py:196-216)
    batch, indices = buf.sample(0)
    assert len(indices) == n_env * 32 and batch.obs.agent_id.dtype == object
    # agents take turns inside every sub-buffer; obs_next carries the NEXT agent's id (quirk Q2)
    first = batch.obs.agent_id[:6].tolist()
    assert first == ["agent_0", "agent_1", "agent_2"] * 2
    assert batch.obs_next.agent_id[:5].tolist() == ["agent_1", "agent_2", "agent_0", "agent_1", "agent_2"]
    assert batch.obs.mask.shape == (n_env * 32, A) and batch.rew.shape == (n_env * 32, N)
    # the reward of a turn is the action taken, credited to the acting agent only
    acting = np.array([env0.agent_idx[a] for a in batch.obs.agent_id])
    assert np.array_equal(batch.rew[np.arange(len(acting)), acting], batch.act.astype(np.float64))
    before = [a.net.flat.data.clone() for a in algos]
    np.random.seed(0)
    with policy_within_training_step(ma):
        stats = ma.update(buf, batch_size=16, repeat=2)
    d = stats.get_loss_stats_dict()
    assert {f"agent_{i}/loss" for i in range(N)} <= set(d) and all(np.isfinite(v) for v in d.values())
    rows_per_agent = [int((acting == i).sum()) for i in range(N)]
    assert d["agent_0/gradient_steps"] == 2 * (rows_per_agent[0] // 16)
    assert all(not torch.equal(b, a.net.flat.data) for b, a in zip(before, algos))
