"""GPU tests (`-m gpu`) of the generic fully-connected engine (csrc/dense.hip) and the CTDE update built on it.

Forward / dgrad / wgrad are compared with a float64 torch autograd replica of the same network (tolerance 1e-5
relative to the tensor's scale, the north star's bar); the CTDE step is compared with the reference's own outputs
(tests/golden/ctde.npz: logits, both losses, every gradient, every post-Adam weight)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.algorithm.multiagent import (
        CentralizedCritic,
        CTDEPolicy,
        DecentralizedActor,
        GlobalStateConstructor,
        SimultaneousTrainer,
        FlexibleMultiAgentPolicyManager,
        agent_batches_from_buffer,
    )
    from tianshou_marl_amd.data import Batch
    from tianshou_marl_amd.utils.net import FlatAdam, FlatMLP

DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _torch_replica(net, dtype=torch.float64):
    layers = []
    for i in range(net.n_layers):
        lin = torch.nn.Linear(net.dims[i], net.dims[i + 1]).to(dtype)
        with torch.no_grad():
            lin.weight.copy_(net.weight(i).cpu().to(dtype))
            lin.bias.copy_(net.bias(i).cpu().to(dtype))
        layers.append(lin)
        if i + 1 < net.n_layers:
            layers.append({"relu": torch.nn.ReLU(), "tanh": torch.nn.Tanh()}[net.act])
    return torch.nn.Sequential(*layers)


def _close(got, want, tol=1e-5):
    want = want.detach().cpu().to(torch.float64)
    scale = max(float(want.abs().max()), 1e-30)
    err = float((got.cpu().to(torch.float64) - want).abs().max())
    assert err <= tol * scale, f"max abs err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("dims,act,B,n_split", [
    ([6, 16, 16, 5], "relu", 32, 1),          # the CTDE fixture's actor
    ([18, 16, 16, 3], "relu", 32, 2),
    ([18, 64, 64, 5], "tanh", 100, 0),        # the fused kernels' shape, through the generic path
    ([48, 128, 128, 5], "relu", 777, 3),      # C3 actor (ragged batch, K not a multiple of 16)
    ([384, 128, 128, 8], "relu", 1030, 5),    # C3 centralized critic: 8 agents x 48
    ([7, 33, 5], "tanh", 65, 2),              # odd everything, 2 layers
    ([5, 3], "relu", 1, 1),                   # single linear layer, single row
    ([130, 257, 70, 66, 2], "relu", 300, 4),  # 4 layers, dims straddling tile edges
])
def test_mlp_forward_backward_match_float64_autograd(dims, act, B, n_split):
    torch.manual_seed(sum(dims) + B)
    net = FlatMLP(dims, act=act, device=DEV, seed=1)
    x = torch.randn(B, dims[0], device=DEV)
    d_out = torch.randn(B, dims[-1], device=DEV)
    out = net(x)
    ref = _torch_replica(net)
    xr = x.cpu().double()
    out_ref = ref(xr)
    _close(out, out_ref.detach())
    slabs = net.backward(d_out, n_split=n_split)
    assert slabs.shape[1] == net.flat.numel() and (n_split == 0 or slabs.shape[0] == n_split)
    (out_ref * d_out.cpu().double()).sum().backward()
    grad = ops.reduce_slabs(slabs)
    lins = [m for m in ref if isinstance(m, torch.nn.Linear)]
    for (gw, gb), lin in zip(net.layer_views(grad), lins):
        _close(gw, lin.weight.grad)
        _close(gb, lin.bias.grad)
    # the forward output is reproducible and independent of what else is in the batch (row-local)
    out2 = net(x[: max(1, B // 2)], save=False)
    assert torch.equal(out2, out[: max(1, B // 2)])


def test_mlp_large_batch_and_descriptor_checks():
    net = FlatMLP([18, 64, 64, 1], act="tanh", device=DEV, seed=2)
    x = torch.randn(76800, 18, device=DEV)  # bench-sized batch: 1200 row tiles
    out = net(x, save=False)
    _close(out, _torch_replica(net)(x.cpu().double()).detach())
    with pytest.raises(ValueError, match="input width"):
        net(torch.zeros(4, 17, device=DEV))
    with pytest.raises(ValueError):
        ops.mlp_desc([4] * 11)
    with pytest.raises(RuntimeError, match="before forward"):
        FlatMLP([4, 2], device=DEV).backward(torch.zeros(1, 2, device=DEV))


def test_flat_adam_matches_torch_adam():
    net = FlatMLP([6, 16, 4], device=DEV, seed=3)
    ref = _torch_replica(net, torch.float32)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=2e-3)
    opt = FlatAdam(net, lr=2e-3)
    x = torch.randn(50, 6, device=DEV)
    for step in range(3):
        d = torch.randn(50, 4, device=DEV)
        net(x)
        opt.step(net.backward(d))
        opt_ref.zero_grad()
        (ref(x.cpu()) * d.cpu()).sum().backward()
        opt_ref.step()
    lins = [m for m in ref if isinstance(m, torch.nn.Linear)]
    for i, lin in enumerate(lins):
        np.testing.assert_allclose(net.weight(i).cpu().numpy(), lin.weight.detach().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(net.bias(i).cpu().numpy(), lin.bias.detach().numpy(), rtol=2e-5, atol=2e-6)


def _fixture_policy(g):
    actor = DecentralizedActor(6, 5, hidden_dim=16, device=DEV)
    critic = CentralizedCritic(18, 3, hidden_dim=16, device=DEV)
    actor.load_layers([(g[f"actor_w{i}"], g[f"actor_b{i}"]) for i in range(3)])
    critic.load_layers([(g[f"critic_w{i}"], g[f"critic_b{i}"]) for i in range(3)])
    return CTDEPolicy(actor=actor, critic=critic, optim_actor=FlatAdam(actor, lr=1e-3), optim_critic=FlatAdam(critic, lr=1e-3),
                      discount_factor=0.99), actor, critic


def test_ctde_learn_matches_reference_fixture():
    g = np.load(os.path.join(GOLD, "ctde.npz"))
    pol, actor, critic = _fixture_policy(g)
    fwd = pol(Batch(obs=g["b_obs"]))
    np.testing.assert_allclose(fwd.act.cpu().numpy(), g["fwd_act_logits"], rtol=1e-5, atol=1e-6)
    batch = Batch(obs=g["b_obs"], act=g["b_act"], rew=g["b_rew"], obs_next=g["b_obs_next"], terminated=g["b_terminated"],
                  global_obs=g["b_global_obs"], global_obs_next=g["b_global_obs_next"])
    # gradients first (same inputs, before the optimizer moves anything)
    obs = torch.as_tensor(g["b_obs"]).to(DEV)
    q_next = critic(torch.as_tensor(g["b_global_obs_next"]).to(DEV), save=False)
    q = critic(torch.as_tensor(g["b_global_obs"]).to(DEV))
    logits = FlatMLP.forward(actor, obs)
    dq, dlogits, scal = ops.ctde_td_head(q, q_next, torch.as_tensor(g["b_rew"]).to(DEV),
                                         torch.as_tensor(g["b_terminated"]).to(DEV, torch.uint8), 0.99, logits,
                                         torch.as_tensor(g["b_act"]).to(DEV))
    np.testing.assert_allclose(scal.cpu().numpy(), [g["actor_loss"], g["critic_loss"]], rtol=1e-5)
    for name, net, d in (("critic", critic, dq), ("actor", actor, dlogits)):
        grad = ops.reduce_slabs(net.backward(d))
        for i, (gw, gb) in enumerate(net.layer_views(grad)):
            ref_w, ref_b = g[f"grad_{name}_w{i}"], g[f"grad_{name}_b{i}"]
            scale = max(np.abs(ref_w).max(), 1e-12)
            assert np.abs(gw.cpu().numpy() - ref_w).max() <= 1e-5 * scale, (name, i)
            assert np.abs(gb.cpu().numpy() - ref_b).max() <= 1e-5 * max(np.abs(ref_b).max(), 1e-12), (name, i)
    # the whole step: losses and post-Adam weights (Adam's first step is +-lr * sign-ish; compare absolutely)
    losses = pol.learn(batch)
    assert losses["actor_loss"] == pytest.approx(float(g["actor_loss"]), rel=1e-5)
    assert losses["critic_loss"] == pytest.approx(float(g["critic_loss"]), rel=1e-5)
    for name, net in (("actor", actor), ("critic", critic)):
        for i in range(3):
            np.testing.assert_allclose(net.weight(i).cpu().numpy(), g[f"after_{name}_w{i}"], rtol=1e-5, atol=2e-6)
            np.testing.assert_allclose(net.bias(i).cpu().numpy(), g[f"after_{name}_b{i}"], rtol=1e-5, atol=2e-6)


def test_ctde_head_quirk_q7_and_local_fallback():
    """actor_loss = -mean(logp) * mean(adv) (the (B,) x (B,1) broadcast), n_out == 1 critics, local-obs fallback."""
    torch.manual_seed(0)
    B, A = 300, 4
    q, qn = torch.randn(B, 1, device=DEV), torch.randn(B, 1, device=DEV)
    rew, term = torch.randn(B, device=DEV), (torch.rand(B, device=DEV) < 0.2).to(torch.uint8)
    logits, act = torch.randn(B, A, device=DEV), torch.randint(0, A, (B,), device=DEV)
    dq, dl, s = ops.ctde_td_head(q, qn, rew, term, 0.9, logits, act)
    qd, ld = q.double().cpu().requires_grad_(), logits.double().cpu().requires_grad_()
    td = rew.double().cpu().unsqueeze(-1) + 0.9 * qn.double().cpu() * (1 - term.double().cpu().unsqueeze(-1))
    critic_loss = torch.nn.functional.mse_loss(qd, td)
    adv = (td - qd).detach()
    logp = -torch.nn.functional.cross_entropy(ld, act.cpu(), reduction="none")
    actor_loss = -(logp * adv).mean()  # (B,) * (B,1) -> (B,B)
    critic_loss.backward()
    actor_loss.backward()
    np.testing.assert_allclose(s.cpu().numpy(), [actor_loss.item(), critic_loss.item()], rtol=1e-5)
    _close(dq, qd.grad)
    _close(dl, ld.grad)
    # no global_obs in the batch -> the critic sees the local observation (ctde.py:144-147)
    actor, critic = DecentralizedActor(6, A, 32, device=DEV, seed=1), CentralizedCritic(6, 1, 32, device=DEV, seed=2)
    pol = CTDEPolicy(actor=actor, critic=critic)
    before = critic.flat.data.clone()
    out = pol.learn(Batch(obs=np.random.randn(40, 6).astype(np.float32), act=np.random.randint(0, A, 40),
                          rew=np.random.randn(40).astype(np.float32), obs_next=np.random.randn(40, 6).astype(np.float32),
                          terminated=np.zeros(40, bool)))
    assert np.isfinite(out["actor_loss"]) and np.isfinite(out["critic_loss"]) and not torch.equal(before, critic.flat.data)
    with pytest.raises(TypeError):
        CTDEPolicy(actor=torch.nn.Linear(2, 2), critic=critic)
    sd = pol.state_dict()
    critic.flat.data.zero_()
    pol.load_state_dict(sd)
    assert not torch.equal(critic.flat.data, torch.zeros_like(critic.flat.data))


def test_global_state_constructor_modes():
    g = np.load(os.path.join(GOLD, "ctde.npz"))
    obs = {f"agent_{i}": torch.as_tensor(g["obs_by_agent"][i]) for i in range(3)}
    np.testing.assert_array_equal(GlobalStateConstructor("concatenate").build(obs).cpu().numpy(), g["global_concatenate"])
    np.testing.assert_allclose(GlobalStateConstructor("mean", obs_dim=6, n_agents=3).build(obs).cpu().numpy(),
                               g["global_mean"], rtol=1e-6, atol=1e-7)
    joint = torch.as_tensor(g["obs_by_agent"]).to(DEV).permute(1, 0, 2).contiguous()  # [B, N, D]
    np.testing.assert_array_equal(GlobalStateConstructor.from_joint_rows(joint).cpu().numpy(), g["global_concatenate"])
    assert GlobalStateConstructor("custom", custom_fn=lambda o: 7).build(obs) == 7
    with pytest.raises(NotImplementedError):
        GlobalStateConstructor("attention", obs_dim=6, n_agents=3)


def test_ctde_policies_train_from_the_device_buffer():
    """Collector (device rollout) -> agent batches with global state -> SimultaneousTrainer -> CTDEPolicy.learn."""
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

    n_env, N, T = 16, 3, 25
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=1)
    behaviour = PPO(net=DiscreteActorCritic(env.obs_dim, 5, 64, device=DEV, seed=1))
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=DEV)
    col = Collector(behaviour, env, buf)
    col.reset()
    with policy_within_training_step(behaviour):
        col.collect(n_step=n_env * T)
    pols = {a: CTDEPolicy(actor=DecentralizedActor(env.obs_dim, 5, 128, device=DEV, seed=i),
                          critic=CentralizedCritic(N * env.obs_dim, N, 128, device=DEV, seed=10 + i),
                          discount_factor=0.0)  # no bootstrapping: the critic regresses the reward, loss must fall
            for i, a in enumerate(env.agents)}
    mgr = FlexibleMultiAgentPolicyManager(pols, env, mode="independent")
    tr = SimultaneousTrainer(mgr)
    batch = agent_batches_from_buffer(buf, env.agents)
    first = tr.train_step(batch)
    for _ in range(30):
        last = tr.train_step(batch)
    assert set(first) == set(env.agents)
    for a in env.agents:  # the centralized critic's regression error shrinks on a fixed batch
        assert last[a]["critic_loss"] < first[a]["critic_loss"]


@pytest.mark.parametrize("n_env", [64, 4096])
def test_c3_ctde_pipeline_full_size(n_env):
    """BASELINE configs[2]: simple_spread N=8 (obs 48), shared decentralized actor + centralized critic on the 384-wide
    concatenated global state, num_envs=4096.  Size-independent properties at full size: episode bookkeeping of the
    rollout, global-state rows, dense forward vs float64 on a row sample, loss head vs a float64 restatement, and a
    learn() step that changes both networks."""
    from tianshou_marl_amd.algorithm.ppo import policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv

    N, T = 8, 25
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=4)
    D = env.obs_dim
    assert D == 48
    pol = CTDEPolicy(actor=DecentralizedActor(D, 5, 128, device=DEV, seed=1),
                     critic=CentralizedCritic(N * D, N, 128, device=DEV, seed=2), seed=9)
    mgr = FlexibleMultiAgentPolicyManager(pol, env, mode="shared")
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, D, device=DEV)
    col = Collector(mgr, env, buf)
    col.reset()
    with policy_within_training_step(mgr):
        st = col.collect(n_step=n_env * T)
        st2_buf_len = len(buf)
    assert st.n_collected_episodes == n_env and (st.lens == T).all() and st2_buf_len == n_env * T
    assert st.returns.shape == (n_env, N) and np.isfinite(st.returns).all()
    # sampled actions follow the actor's distribution: stored logp == log_softmax(actor(obs))[act]
    obs_all = buf.obs_store[:T].reshape(-1, D)
    logits = FlatMLP.forward(pol.actor, obs_all, save=False)
    ref_lp = torch.log_softmax(logits.double(), -1).gather(1, buf.act_store[:T].reshape(-1, 1).long()).squeeze(1)
    assert float((buf.logp_store[:T].reshape(-1).double() - ref_lp).abs().max()) < 1e-5
    counts = torch.bincount(buf.act_store[:T].reshape(-1).long(), minlength=5).double()
    expect = torch.softmax(logits.double(), -1).sum(0)
    assert float(((counts - expect) ** 2 / expect).sum()) < 40.0  # chi^2, 4 dof
    batch = agent_batches_from_buffer(buf, env.agents)
    R = n_env * T
    assert batch.global_obs.shape == (R, N * D) and batch.agent_3.obs.shape == (R, D)
    # the concatenated global state of a row holds every agent's observation of that joint step, in agent order
    assert torch.equal(batch.global_obs[:, 3 * D:4 * D], batch.agent_3.obs)
    # dense forward of the 384-wide critic vs float64 on a row sample
    idx = torch.randint(0, R, (2048,), device=DEV)
    q_all = pol.critic(batch.global_obs, save=False)
    ref = _torch_replica(pol.critic)(batch.global_obs[idx].cpu().double())
    _close(q_all[idx], ref.detach())
    # loss head at full size vs float64 (same formulas as ctde.py:149-185)
    ab = batch.agent_0
    q_next = pol.critic(batch.global_obs_next, save=False)
    lg = FlatMLP.forward(pol.actor, ab.obs, save=False)
    dq, dl, s = ops.ctde_td_head(q_all, q_next, ab.rew, ab.terminated.to(torch.uint8), 0.99, lg, ab.act)
    v, vn = q_all.double().mean(1), q_next.double().mean(1)
    td = ab.rew.double() + 0.99 * vn * (1 - ab.terminated.double())
    lp = torch.log_softmax(lg.double(), -1).gather(1, ab.act.view(-1, 1)).squeeze(1)
    assert float(s[1]) == pytest.approx(float(((v - td) ** 2).mean()), rel=1e-5)
    assert float(s[0]) == pytest.approx(float(-lp.mean() * (td - v).mean()), rel=1e-4, abs=1e-7)
    _close(dq, (2 * (v - td) / (R * N)).unsqueeze(1).expand(R, N))
    # one learn() step per agent batch on the shared policy moves both networks and returns finite losses
    before = (pol.actor.flat.data.clone(), pol.critic.flat.data.clone())
    from tianshou_marl_amd.algorithm.multiagent.training_coordinator import _attach_global

    assert pol._store_path(_attach_global(batch, batch.agent_0)) is not None  # the one-launch kernels on the stores serve this shape
    out = SimultaneousTrainer(mgr).train_step(batch)
    assert set(out) == set(env.agents) and all(np.isfinite(list(v.values())).all() for v in out.values())
    assert not torch.equal(before[0], pol.actor.flat.data) and not torch.equal(before[1], pol.critic.flat.data)
    # the first learner (agent_0) saw the initial weights: its losses, from the fused kernels reading the stores in place at
    # FULL size, equal the float64 restatement above (and the dense head's `s`)
    assert float(out["agent_0"]["critic_loss"]) == pytest.approx(float(((v - td) ** 2).mean()), rel=2e-5)
    assert float(out["agent_0"]["actor_loss"]) == pytest.approx(float(-lp.mean() * (td - v).mean()), rel=1e-4, abs=1e-6)


@pytest.mark.parametrize("max_cycles,T", [(6, 6), (4, 6)])
def test_ctde_learn_takes_next_values_from_the_chained_forward(max_cycles, T):
    """CTDEPolicy.learn evaluates the ONLINE critic on obs_next for its TD target (ctde.py:165-172); on per-agent batches of
    chained rows (agent_batches_from_buffer after a Collector) those values are rows of its pass over obs, plus the last
    step of every env block -- or the full pass when an episode ended early (device flag).  Losses, gradients' effect on
    the weights and the optimizer state must equal the two-pass form bit for bit."""
    from tianshou_marl_amd.algorithm.multiagent import (CentralizedCritic, CTDEPolicy, DecentralizedActor,
                                                        FlexibleMultiAgentPolicyManager, SimultaneousTrainer,
                                                        agent_batches_from_buffer)
    from tianshou_marl_amd.algorithm.ppo import policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv

    n_env, N = 40, 3
    outs = []
    for chained in (True, False):
        env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=max_cycles, device=DEV, seed=9)
        D = env.obs_dim
        # (fused=False: the dense-GEMM path with tsm_ctde_td_head; the fused path on the stores has its own test below)
        pol = CTDEPolicy(actor=DecentralizedActor(D, 5, 128, device=DEV, seed=1),
                         critic=CentralizedCritic(N * D, N, 128, device=DEV, seed=2), seed=4, fused=False)
        mgr = FlexibleMultiAgentPolicyManager(pol, env, mode="shared")
        buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, D, device=DEV)
        col = Collector(mgr, env, buf)
        col.reset()
        trainer = SimultaneousTrainer(mgr)
        losses = []
        for _ in range(2):
            with policy_within_training_step(mgr):
                col.collect(n_step=n_env * T)
                batch = agent_batches_from_buffer(buf, env.agents)
                assert "chain_done" in batch and batch.chain_done.chain_T == T
                if not chained:
                    batch.pop("chain_done")
                losses.append(trainer.train_step(batch))
            col.reset_buffer(keep_statistics=True)
        outs.append((pol.actor.flat.data.clone(), pol.critic.flat.data.clone(), pol.optim_critic.exp_avg_sq.clone(), losses))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[3] == b[3]


@pytest.mark.parametrize("n_env,max_cycles,rounds,graph", [(40, 25, 4, True), (600, 25, 2, True), (40, 7, 4, True), (40, 25, 2, False)])
def test_ctde_learn_on_the_stores_equals_learn_on_the_copies(n_env, max_cycles, rounds, graph):
    """CTDEPolicy.learn (ctde.py:121-199) two ways on the SAME collected rows of BASELINE configs[2]'s shape: the
    unfused path on env-major copies (dense GEMMs + tsm_ctde_td_head, pinned to the reference by ctde.npz) and the fused
    path that reads the time-major stores in place (csrc/critic_rows.hip, critic_train.hip, critic_dw1.hip, ppo_rows.hip,
    tsm_ctde_finalize, tsm_adam_step_segs).  Losses and both networks' post-update weights agree over two rounds of all
    eight agents' updates (the summation orders differ: 2e-5 of the weight scale for all but a few ill-conditioned
    Adam quotients).  max_cycles 7: episodes end in the middle of the 25 collected slots, where obs_next of a row is not
    the next slot's obs -- a device flag switches the targets to a full V(obs_next) pass, no host round trip.  graph: the
    fused learn replays as ONE hipGraph per call from its second call on (two graphs per agent take turns: four rounds
    replay both)."""
    from tianshou_marl_amd.algorithm.ppo import policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv

    N, T = 8, 25
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=max_cycles, device=DEV, seed=4)
    D = env.obs_dim
    mk = lambda fused: CTDEPolicy(actor=DecentralizedActor(D, 5, 128, device=DEV, seed=1),  # noqa: E731
                                  critic=CentralizedCritic(N * D, N, 128, device=DEV, seed=2), seed=9, fused=fused, graph=graph)
    pol_f, pol_u = mk(True), mk(False)
    actor0 = pol_f.actor.flat.data.double().cpu().numpy()
    mgr = FlexibleMultiAgentPolicyManager(pol_f, env, mode="shared")
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, D, device=DEV)
    col = Collector(mgr, env, buf)
    col.reset()
    for rnd in range(rounds):
        with policy_within_training_step(mgr):
            col.collect(n_step=n_env * T)
        lazy = agent_batches_from_buffer(buf, env.agents, copies=False)
        full = agent_batches_from_buffer(buf, env.agents)
        assert "obs" not in lazy.agent_0 and getattr(lazy.chain_done, "store", None) is not None
        assert int(full.agent_5.agent_index) == 5 and full.chain_done.store.T == T
        assert int(lazy.chain_done.store.early_done.item()) == int(max_cycles < T)
        for a in env.agents:
            from tianshou_marl_amd.algorithm.multiagent.training_coordinator import _attach_global

            assert pol_f._store_path(_attach_global(lazy, lazy[a])) is not None
            lf = pol_f.learn(_attach_global(lazy, lazy[a]))
            lu = pol_u.learn(_attach_global(full, full[a]))
            assert lf["critic_loss"] == pytest.approx(lu["critic_loss"], rel=2e-5, abs=1e-7), a
            assert lf["actor_loss"] == pytest.approx(lu["actor_loss"], rel=2e-5, abs=1e-7), a
            if rnd == 0 and a == env.agents[1]:  # two updates in: every actor weight still agrees tightly
                np.testing.assert_allclose(pol_f.actor.flat.data.cpu().numpy(), pol_u.actor.flat.data.cpu().numpy(),
                                           rtol=1e-4, atol=1e-5)
        col.reset_buffer(keep_statistics=True)
    # critic: every weight, tightly
    pf, pu = pol_f.critic.flat.data.double().cpu().numpy(), pol_u.critic.flat.data.double().cpu().numpy()
    np.testing.assert_allclose(pf, pu, rtol=1e-4, atol=1e-4 * np.abs(pu).max())
    # actor: its gradient is mean(advantage) x the score function summed over the rows (quirk Q7) -- zero-mean noise whose
    # entries are mostly ~1e-6 and smaller, which both paths get right to f32 summation error (1e-6 of the gradient norm, the
    # losses above agree to six digits) but Adam's g / (|g| + 1e-8) turns a 1e-8 difference on such an entry into percents of
    # a step.  So: tight after the first two updates (checked above through the snapshot), and the whole movement of the
    # two paths points the same way
    af, au = pol_f.actor.flat.data.double().cpu().numpy(), pol_u.actor.flat.data.double().cpu().numpy()
    mf, mu = af - actor0, au - actor0
    assert float(mf @ mu / (np.linalg.norm(mf) * np.linalg.norm(mu))) > 0.999
    assert np.abs(af - au).max() <= 2 * 8 * rounds * pol_u.optim_actor.lr
    if graph:
        assert sum(len(w["graphs"]) for w in pol_f._ws.values() if isinstance(w, dict) and "graphs" in w) == 8 * min(2, rounds - 1)
    assert pol_f.optim_actor.step_count == pol_u.optim_actor.step_count == pol_f.optim_critic.step_count == 8 * rounds


def _wide_flat(g, prefix, name):
    """Layer tensors of the fixture in FlatMLP's flat order (per layer: weight, bias)."""
    return np.concatenate([np.concatenate([g[f"{prefix}_{name}_w{i}"].reshape(-1), g[f"{prefix}_{name}_b{i}"].reshape(-1)])
                           for i in range(3)]).astype(np.float64)


@pytest.mark.parametrize("tile", [32, 64], ids=["tile32", "tile64"])
@pytest.mark.parametrize("path", ["store", "store_eager", "copies"])
@pytest.mark.parametrize("variant", ["chain", "early"])
@pytest.mark.parametrize("fixture", ["ctde_wide.npz", "ctde_c3.npz"])
def test_ctde_learn_wide_matches_reference_fixture(fixture, variant, path, tile):
    """CTDEPolicy.learn against the REFERENCE's own run with 128-wide nets (128-wide actor, centralized critic on the N*D joint
    row, every agent's learn() call in turn on the same rows, as the MARL trainers issue them).  `ctde_wide.npz`: N = 4,
    D = 24 -- critic input K1 = 96, the <6> instantiations of the critic kernels; `ctde_c3.npz`: BASELINE configs[2]'s own
    widths, N = 8, D = 48 -- K1 = 384, 8 outputs: the <24> instantiations of critic_rows_train_kernel / critic_dw1_kernel /
    critic_rows_forward_kernel that the bench runs.  `tile`: the actor gradient step on the 32-sample kernel (csrc/ppo_rows.hip)
    and on the 64-sample kernel (csrc/actor_rows64.hip), forced through the `actor_tile` kernel option.  `store`: the one-launch kernels reading the time-major stores in place (critic_rows / critic_train / critic_dw1 /
    ppo_rows / ctde_finalize / adam_step_segs; the 2nd call of an agent would replay a hipGraph -- here every agent is called
    once, so `store` and `store_eager` differ in nothing but the flag and both must hold); `copies`: dense GEMMs on env-major
    copies.  `early`: episodes end mid-store, obs_next of those rows is not the next slot's obs.
    Bars: both losses of every call 1e-5; weights after the first call rtol 1e-5 + atol 5e-6 plus what ONE Adam step from zero
    moments does to a gradient known to 1e-5 of its scale (u = lr g / (|g| + eps): du = lr eps dg / (|g| + eps)^2, at most a
    step of lr where |g| < dg); after all N calls every weight of both nets rtol 1e-5 + atol 5e-6 plus the same allowance
    summed over the calls from the reference's own Adam state (`*_adamcond_*`), and 99.9 % of them without it.  (The generator picks
    data whose every ReLU pre-activation in the reference's run stays 2e-6 off the kink -- `*_min_abs_preact` in the fixture:
    a unit at |z| ~ 1e-7 takes either side depending on the f32 summation order and Adam turns that into +-lr steps.)"""
    from tianshou_marl_amd.algorithm.multiagent.training_coordinator import _attach_global
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer

    if path == "copies" and tile == 64:
        pytest.skip("the dense path on copies has no actor tile")
    g = np.load(os.path.join(GOLD, fixture))
    N, D, A, H, E, T = (int(x) for x in g["dims"])
    lr, eps = float(g["lr"]), 1e-8
    with ops.kernel_override(actor_tile=tile):
        _ctde_wide_replay(g, variant, path, N, D, A, H, E, T, lr, eps)
    assert ops.kernel_option("actor_tile") == 0


def _ctde_wide_replay(g, variant, path, N, D, A, H, E, T, lr, eps):
    from tianshou_marl_amd.algorithm.multiagent.training_coordinator import _attach_global
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer

    actor = DecentralizedActor(D, A, hidden_dim=H, device=DEV)
    critic = CentralizedCritic(N * D, N, hidden_dim=H, device=DEV)
    actor.load_layers([(g[f"{variant}_init_actor_w{i}"], g[f"{variant}_init_actor_b{i}"]) for i in range(3)])
    critic.load_layers([(g[f"{variant}_init_critic_w{i}"], g[f"{variant}_init_critic_b{i}"]) for i in range(3)])
    pol = CTDEPolicy(actor=actor, critic=critic, optim_actor=FlatAdam(actor, lr=lr), optim_critic=FlatAdam(critic, lr=lr),
                     discount_factor=float(g["gamma"]), fused=path != "copies", graph=path == "store")
    buf = DeviceVectorReplayBuffer(E * T, E, N, D, device=DEV)
    f = lambda k, t: g[f"{variant}_{k}"][t]  # noqa: E731
    for t in range(T):
        buf.add(Batch(obs=f("obs", t), act=f("act", t), rew=f("rew", t), terminated=f("term", t), truncated=f("trunc", t),
                      obs_next=f("obs_next", t)))
    buf.mark_rows_chained("empty", True)  # the fixture's rows continue one another by construction (what a Collector marks)
    agents = [f"agent_{i}" for i in range(N)]
    batches = agent_batches_from_buffer(buf, agents, copies=path == "copies")
    for a, name in enumerate(agents):
        b = _attach_global(batches, batches[name])
        assert (pol._store_path(b) is not None) == (path != "copies")
        out = pol.learn(b)
        ref_al, ref_cl = (float(x) for x in g[f"{variant}_losses"][a])
        assert out["critic_loss"] == pytest.approx(ref_cl, rel=1e-5), (a, "critic")
        assert out["actor_loss"] == pytest.approx(ref_al, rel=1e-5, abs=1e-7), (a, "actor")
        if a == 0:
            for nm, net in (("actor", actor), ("critic", critic)):
                ref, gr = _wide_flat(g, f"{variant}_after1", nm), _wide_flat(g, f"{variant}_grad1", nm)
                dg = 1e-5 * np.abs(gr).max()
                tol = 5e-6 + 1e-5 * np.abs(ref) + np.minimum(lr * eps * dg / (np.abs(gr) + eps) ** 2, 2 * lr)
                got = net.flat.data.double().cpu().numpy()
                bad = np.abs(got - ref) > tol
                assert not bad.any(), (nm, int(bad.sum()), float(np.abs(got - ref)[bad].max()))
                # and most of the weights are nowhere near those ill-conditioned entries: plain tolerance for 95 % of them
                assert (np.abs(got - ref) <= 5e-6 + 1e-5 * np.abs(ref)).mean() > 0.95, nm
    for nm, net in (("actor", actor), ("critic", critic)):  # N Adam steps in: every weight of both nets
        ref, got = _wide_flat(g, f"{variant}_afterN", nm), net.flat.data.double().cpu().numpy()
        # plain tolerance + what N Adam steps do to a gradient known to 1e-5 of its scale: `adamcond` is the reference's own
        # sum over the calls of lr / (sqrt(v^_k) + eps) per parameter (make_fixtures.py), a handful of parameters whose
        # gradient all but cancelled in the first call (|g| ~ 1e-9: a sign-like step of up to lr) carry it to the end
        dg = 1e-5 * np.abs(_wide_flat(g, f"{variant}_grad1", nm)).max()
        tol = 5e-6 + 1e-5 * np.abs(ref) + np.minimum(_wide_flat(g, f"{variant}_adamcond", nm) * dg, 2 * lr * N)
        bad = np.abs(got - ref) > tol
        assert not bad.any(), (nm, int(bad.sum()), float(np.abs(got - ref)[bad].max()))
        assert (np.abs(got - ref) <= 5e-6 + 1e-5 * np.abs(ref)).mean() > 0.999, nm  # ... and all but a handful: plain tolerance
    assert pol.optim_actor.step_count == pol.optim_critic.step_count == N
