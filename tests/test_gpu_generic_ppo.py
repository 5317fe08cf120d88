"""GPU tests (`-m gpu`) of GenericPPO (PPO on actor / critic MLPs of any width, optional centralized critic): one full
gradient step against a float64 torch-autograd replica of the reference's arithmetic (a2c.py:113-151, ppo.py:164-224,
clip_grad_norm_ + Adam over actor + critic), and the rollout -> update pipeline on the device env."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))

if torch.cuda.is_available():
    from tianshou_marl_amd.algorithm import GenericPPO, policy_within_training_step
    from tianshou_marl_amd.data import Batch
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.utils.net import MLPActorCritic

DEV = "cuda"


def _replica(net):
    def mlp(f):
        layers = []
        for i in range(f.n_layers):
            lin = torch.nn.Linear(f.dims[i], f.dims[i + 1]).double()
            with torch.no_grad():
                lin.weight.copy_(f.weight(i).cpu().double())
                lin.bias.copy_(f.bias(i).cpu().double())
            layers.append(lin)
            if i + 1 < f.n_layers:
                layers.append(torch.nn.Tanh() if f.act == "tanh" else torch.nn.ReLU())
        return torch.nn.Sequential(*layers)

    return mlp(net.actor), mlp(net.critic)


@pytest.mark.parametrize("hidden,act,glob,opts", [
    ((32, 32), "tanh", False, {}),
    ((128, 128), "relu", True, dict(max_grad_norm=0.5, value_clip=True)),
    ((48,), "relu", False, dict(dual_clip=3.0, advantage_normalization=False)),
])
def test_generic_ppo_step_matches_float64_autograd(oracle, hidden, act, glob, opts):
    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    n, D, A, N = 300, 10, 4, 3
    net = MLPActorCritic(D, A, hidden, act=act, critic_obs_dim=N * D if glob else None, device=DEV, seed=5)
    algo = GenericPPO(net=net, lr=1e-3, critic_input="global" if glob else "local", n_agent=N if glob else 1,
                      shuffle="numpy", dispatch="pooled", **opts)
    obs, obs_next = rng.standard_normal((n, D)).astype(np.float32), rng.standard_normal((n, D)).astype(np.float32)
    g, g_next = rng.standard_normal((n, N * D)).astype(np.float32), rng.standard_normal((n, N * D)).astype(np.float32)
    act_np = rng.integers(0, A, n)
    rew = rng.standard_normal(n).astype(np.float32)
    term = rng.random(n) < 0.05
    trunc = np.zeros(n, bool)
    batch = Batch(obs=obs, act=act_np, rew=rew, obs_next=obs_next, terminated=term, truncated=trunc)
    if glob:
        batch.global_obs, batch.global_obs_next = g, g_next
    actor, critic = _replica(net)
    params = list(actor.parameters()) + list(critic.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)
    # ---- float64 replica of one full-batch PPO step ----
    to = lambda x: torch.as_tensor(x).double()  # noqa: E731
    cin, cin_next = (to(g), to(g_next)) if glob else (to(obs), to(obs_next))
    with torch.no_grad():
        v_s, v_next = critic(cin).flatten(), critic(cin_next).flatten()
        logp_old = torch.log_softmax(actor(to(obs)), -1).gather(1, torch.as_tensor(act_np).view(-1, 1)).flatten()
    ret, adv = oracle.gae_lanes(v_s.numpy().astype(np.float32).reshape(n, 1), v_next.numpy().astype(np.float32).reshape(n, 1),
                                rew.reshape(n, 1), term.reshape(n, 1), trunc.reshape(n, 1), 0.99, 0.95)
    ret, adv = torch.as_tensor(ret.astype(np.float32)).double().flatten(), torch.as_tensor(adv.astype(np.float32)).double().flatten()
    a = adv
    if opts.get("advantage_normalization", True):
        a = (a - a.mean()) / (a.std() + 1e-8)
    logits = actor(to(obs))
    lsm = torch.log_softmax(logits, -1)
    logp = lsm.gather(1, torch.as_tensor(act_np).view(-1, 1)).flatten()
    ratio = (logp - logp_old).exp()
    s1, s2 = ratio * a, ratio.clamp(0.8, 1.2) * a
    if opts.get("dual_clip"):
        clip1 = torch.min(s1, s2)
        clip2 = torch.max(clip1, opts["dual_clip"] * a)
        clip_loss = -torch.where(a < 0, clip2, clip1).mean()
    else:
        clip_loss = -torch.min(s1, s2).mean()
    value = critic(cin).flatten()
    if opts.get("value_clip"):
        v_clip = v_s + (value - v_s).clamp(-0.2, 0.2)
        vf_loss = torch.max((ret - value) ** 2, (ret - v_clip) ** 2).mean()
    else:
        vf_loss = ((ret - value) ** 2).mean()
    ent = -(lsm.exp() * lsm).sum(-1).mean()
    loss = clip_loss + 0.5 * vf_loss - 0.01 * ent
    opt.zero_grad()
    loss.backward()
    if opts.get("max_grad_norm"):
        torch.nn.utils.clip_grad_norm_(params, opts["max_grad_norm"])
    opt.step()
    # ---- the HIP path ----
    out = algo.learn(batch, batch_size=None, repeat=1)
    assert out["loss"] == pytest.approx(float(loss), rel=2e-5, abs=1e-6)
    assert out["vf_loss"] == pytest.approx(float(vf_loss), rel=2e-5) and out["ent_loss"] == pytest.approx(float(ent), rel=2e-5)
    for f, ref in ((net.actor, actor), (net.critic, critic)):
        lins = [m for m in ref if isinstance(m, torch.nn.Linear)]
        for i, lin in enumerate(lins):
            np.testing.assert_allclose(f.weight(i).cpu().numpy(), lin.weight.detach().numpy(), rtol=2e-5, atol=3e-6)
            np.testing.assert_allclose(f.bias(i).cpu().numpy(), lin.bias.detach().numpy(), rtol=2e-5, atol=3e-6)


@pytest.mark.parametrize("glob,hidden", [(False, (128, 128)), (True, (64, 64))])
def test_generic_ppo_rollout_and_update_on_device_env(glob, hidden):
    n_env, N, T = 32, 3, 25
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=4)
    D = env.obs_dim
    net = MLPActorCritic(D, 5, hidden, critic_obs_dim=N * D if glob else None, device=DEV, seed=1)
    algo = GenericPPO(net=net, critic_input="global" if glob else "local", n_agent=N, shuffle="device", seed=2)
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, D, device=DEV)
    col = Collector(algo, env, buf)
    # a 128-wide actor rolls out in the actor-only persistent kernel (csrc/rollout_rows.hip); the 64-wide MLPActorCritic
    # here does not (the 64-wide fused rollout is for the DiscreteActorCritic layout only)
    assert col._can_fuse() == (hidden == (128, 128))
    col.reset()
    losses = []
    for _ in range(3):
        with policy_within_training_step(algo):
            st = col.collect(n_step=n_env * T)
            # stored behaviour log-probs are those of the actor; a centralized value is shared by an env's agents
            if glob:
                v = buf.vs_store[:T]
                assert torch.equal(v[:, :, 0], v[:, :, 1]) and torch.equal(v[:, :, 0], v[:, :, 2])
            before = net.flat.data.clone()
            ts = algo.update(buf, batch_size=512, repeat=2)
        col.reset_buffer(keep_statistics=True)
        assert st.n_collected_episodes == n_env
        d = ts.get_loss_stats_dict()
        assert all(np.isfinite(v) for v in d.values()) and d["agent_0/gradient_steps"] == 2
        assert not torch.equal(before, net.flat.data)
        losses.append(d["agent_1/vf_loss"])
    # snapshots for opponent pools are independent deep copies
    import copy

    snap = copy.deepcopy(algo)
    assert torch.equal(snap.net.flat.data, net.flat.data) and snap.net.flat.data_ptr() != net.flat.data_ptr()
    assert snap.critic_input == algo.critic_input and snap.opt_step == algo.opt_step


def test_generic_ppo_graph_replay_equals_eager_launches():
    """Capturing the whole update (critic passes, GAE, every gradient step) into one hipGraph changes no bit."""
    def run(graph):
        env = DeviceSimpleSpreadVectorEnv(16, 3, max_cycles=25, device=DEV, seed=6)
        net = MLPActorCritic(env.obs_dim, 5, (96, 96), act="tanh", critic_obs_dim=3 * env.obs_dim, device=DEV, seed=6)
        algo = GenericPPO(net=net, critic_input="global", n_agent=3, shuffle="numpy", seed=6, graph=graph, max_grad_norm=0.5)
        buf = DeviceVectorReplayBuffer(16 * 25, 16, 3, env.obs_dim, device=DEV)
        col = Collector(algo, env, buf, use_graph=False)
        col.reset()
        np.random.seed(9)
        losses = []
        for _ in range(4):  # graph mode: 1st update eager, 2nd captures + replays, 3rd / 4th replay
            with policy_within_training_step(algo):
                col.collect(n_step=16 * 25)
                ts = algo.update(buf, 128, 2)
            col.reset_buffer(keep_statistics=True)
            losses.append(ts.get_loss_stats_dict())
        return net.flat.data.cpu().numpy(), losses, algo.opt_step

    p_e, l_e, s_e = run(False)
    p_g, l_g, s_g = run(True)
    assert s_e == s_g == 4 * 3 * 2 * 3  # 400 rows per agent / 128 -> 3 minibatches, 2 repeats, 3 agents, 4 updates
    assert np.array_equal(p_e, p_g)
    assert l_e == l_g


def test_c3_update_at_full_size_graph_equals_eager():
    """BASELINE configs[2] at its full size (4096 envs x 8 agents x T = 25, minibatch 65 536 as whole joint rows): three
    updates through the captured graph and through eager launches end with the same bits (64-sample actor kernel, two-launch
    critic step, segmented Adam, one-launch V(obs), chained next values, the rollout's log-probabilities reused as logp_old):
    a size-independent property checked at the size the bench runs."""
    def run(graph):
        n_env, N, T = 4096, 8, 25
        env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=11)
        net = MLPActorCritic(env.obs_dim, 5, (128, 128), critic_obs_dim=N * env.obs_dim, device=DEV, seed=11)
        algo = GenericPPO(net=net, critic_input="global", n_agent=N, shuffle="device", seed=11, dispatch="pooled", graph=graph)
        buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=DEV)
        col = Collector(algo, env, buf)
        col.reset()
        losses = []
        for _ in range(3):
            with policy_within_training_step(algo):
                col.collect(n_step=n_env * T)
                ts = algo.update(buf, 65536, 1)
            col.reset_buffer(keep_statistics=True)
            losses.append(ts.get_loss_stats_dict())
        return net.flat.data.clone(), algo.exp_avg_sq.clone(), losses, algo.opt_step

    p_g, v_g, l_g, s_g = run(True)
    p_e, v_e, l_e, s_e = run(False)
    assert s_g == s_e == 3 * 12  # 102 400 joint rows / 8 192 -> 12 minibatches + merge-last
    assert torch.equal(p_g, p_e) and torch.equal(v_g, v_e) and l_g == l_e
    assert all(np.isfinite(list(d.values())).all() for d in l_g)


def test_generic_ppo_async_statistics_equal_the_synchronous_ones():
    """GenericPPO(async_stats=True): update() returns before the device has finished and the training statistics are read one
    step late from a ring of pinned slots (what bench.py --workload c3ppo does) -- the same numbers and the same weights as
    the synchronous read, over more updates than the ring has slots; statistics left unread for four updates expire."""
    def run(async_stats):
        env = DeviceSimpleSpreadVectorEnv(16, 3, max_cycles=25, device=DEV, seed=6)
        net = MLPActorCritic(env.obs_dim, 5, (128, 128), critic_obs_dim=3 * env.obs_dim, device=DEV, seed=6)
        algo = GenericPPO(net=net, critic_input="global", n_agent=3, shuffle="device", seed=6, dispatch="pooled", async_stats=async_stats)
        buf = DeviceVectorReplayBuffer(16 * 25, 16, 3, env.obs_dim, device=DEV)
        col = Collector(algo, env, buf)
        col.reset()
        losses, prev = [], None
        for _ in range(7):
            with policy_within_training_step(algo):
                col.collect(n_step=16 * 25)
                ts = algo.update(buf, 96 * 3, 1)
            if prev is not None:
                losses.append(prev.get_loss_stats_dict())  # one step late
            prev = ts
            col.reset_buffer(keep_statistics=True)
        losses.append(prev.get_loss_stats_dict())
        return net.flat.data.cpu().numpy(), losses, algo

    p_s, l_s, _ = run(False)
    p_a, l_a, algo = run(True)
    assert np.array_equal(p_s, p_a) and l_s == l_a and len(l_a) == 7
    # unread statistics are dropped after four further updates, loudly
    env = DeviceSimpleSpreadVectorEnv(16, 3, max_cycles=25, device=DEV, seed=6)
    buf = DeviceVectorReplayBuffer(16 * 25, 16, 3, env.obs_dim, device=DEV)
    col = Collector(algo, env, buf)
    col.reset()
    held = []
    for _ in range(6):
        with policy_within_training_step(algo):
            col.collect(n_step=16 * 25)
            held.append(algo.update(buf, 96 * 3, 1))
        col.reset_buffer(keep_statistics=True)
    # (this buffer's first update runs eagerly and returns finished statistics; the replays behind it use the ring: the slot of
    # update 1 is taken again by update 5)
    assert np.isfinite(list(held[0].get_loss_stats_dict().values())).all()
    with pytest.raises(RuntimeError, match="not read within 4"):
        held[1].get_loss_stats_dict()
    assert np.isfinite(list(held[-1].get_loss_stats_dict().values())).all()


@pytest.mark.parametrize("N,graph,H", [(3, False, 32), (8, True, 32), (4, False, 32), (8, True, 128), (3, False, 128)])
def test_row_minibatches_with_centralized_critic_match_float64_autograd(oracle, N, graph, H):
    """Pooled dispatch + centralized critic: a minibatch is a set of joint rows; the critic runs once per row on the
    concatenated observations and its value / gradient is shared by the row's N agents (value_group in the loss kernel:
    wave-shuffle sum for N = 4 / 8, one thread per row for N = 3).  Whole update vs a float64 autograd replica that uses
    the same permutations (reference sample(0) order).  H = 128: the actor runs in the one-launch kernel of
    csrc/ppo_rows.hip, the critic beside it with the value term alone."""
    from tianshou_marl_amd.data.batch import split_bounds

    torch.manual_seed(1)
    B, T, D, A = 12, 6, 5, 4
    net = MLPActorCritic(D, A, (H, H), critic_obs_dim=N * D, device=DEV, seed=2)
    gn = None if (H == 128 and not graph) else 0.7  # without clipping each half's slabs feed its own Adam launch
    algo = GenericPPO(net=net, critic_input="global", n_agent=N, dispatch="pooled", shuffle="numpy", lr=1e-3, graph=graph,
                      max_grad_norm=gn, value_clip=True)
    assert algo.row_minibatches and algo.fused_actor == (H == 128)
    buf = DeviceVectorReplayBuffer(B * T, B, N, D, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    for t in range(T):
        r = lambda *s: torch.randn(*s, device=DEV, generator=gen)  # noqa: E731
        term = (torch.rand(B, N, device=DEV, generator=gen) < 0.1).to(torch.uint8)
        buf.add_device(r(B, N, D), torch.randint(0, A, (B, N), device=DEV, generator=gen, dtype=torch.int32), r(B, N), term,
                       torch.full((B, N), int(t == T - 1), dtype=torch.uint8, device=DEV), obs_next=r(B, N, D))
    actor, critic = _replica(net)
    params = list(actor.parameters()) + list(critic.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)
    c = lambda x: x.cpu().numpy()  # noqa: E731
    obs, obs_next = torch.as_tensor(c(buf.obs_store)).double(), torch.as_tensor(c(buf.obs_next_store)).double()   # [T, B, N, D]
    act = torch.as_tensor(c(buf.act_store)).long()
    with torch.no_grad():
        v_s = critic(obs.reshape(T * B, N * D)).reshape(T, B, 1).expand(T, B, N)
        v_n = critic(obs_next.reshape(T * B, N * D)).reshape(T, B, 1).expand(T, B, N)
        logp_old = torch.log_softmax(actor(obs), -1).gather(-1, act.unsqueeze(-1)).squeeze(-1)
    L = B * N
    ret, adv = oracle.gae_lanes(v_s.numpy().astype(np.float32).reshape(T, L), v_n.numpy().astype(np.float32).reshape(T, L),
                                c(buf.rew_store).reshape(T, L), c(buf.term_store).reshape(T, L).astype(bool),
                                c(buf.trunc_store).reshape(T, L).astype(bool), 0.99, 0.95)
    ret = torch.as_tensor(ret.astype(np.float32)).double().reshape(T * B, N)
    adv = torch.as_tensor(adv.astype(np.float32)).double().reshape(T * B, N)
    v_old = v_s.reshape(T * B, N).float().double()
    ref_rows = (np.arange(T)[None, :] * B + np.arange(B)[:, None]).reshape(-1)  # sample(0) order -> joint-row id
    batch_size, repeat = 20 * N, 2
    np.random.seed(4)
    losses = []
    for _ in range(repeat):
        perm = ref_rows[np.random.permutation(T * B)]
        for s, e in split_bounds(T * B, batch_size // N, True):
            rows = torch.as_tensor(perm[s:e])
            x = obs.reshape(T * B, N, D)[rows]
            a = adv[rows]
            a = (a - a.mean()) / (a.std() + 1e-8)
            lsm = torch.log_softmax(actor(x), -1)
            logp = lsm.gather(-1, act.reshape(T * B, N)[rows].unsqueeze(-1)).squeeze(-1)
            ratio = (logp - logp_old.reshape(T * B, N)[rows]).exp()
            clip_loss = -torch.min(ratio * a, ratio.clamp(0.8, 1.2) * a).mean()
            value = critic(x.reshape(len(rows), N * D)).expand(len(rows), N)
            v_clip = v_old[rows] + (value - v_old[rows]).clamp(-0.2, 0.2)
            vf_loss = torch.max((ret[rows] - value) ** 2, (ret[rows] - v_clip) ** 2).mean()
            ent = -(lsm.exp() * lsm).sum(-1).mean()
            loss = clip_loss + 0.5 * vf_loss - 0.01 * ent
            opt.zero_grad()
            loss.backward()
            if gn:
                torch.nn.utils.clip_grad_norm_(params, gn)
            opt.step()
            losses.append(float(loss))
    np.random.seed(4)
    with policy_within_training_step(algo):
        if graph:  # the first update of a shape runs eagerly, the second captures: replay the same update twice on copies
            p0 = net.flat.data.clone()
            algo.update(buf, batch_size, repeat)
            net.flat.data.copy_(p0)
            algo.exp_avg.zero_()
            algo.exp_avg_sq.zero_()
            algo.opt_step = 0
            np.random.seed(4)
        st = algo.update(buf, batch_size, repeat)
    assert st.gradient_steps == len(losses) == 2 * len(split_bounds(T * B, 20, True))
    assert st.loss.mean == pytest.approx(np.mean(losses), rel=2e-5, abs=2e-6)
    for f, ref in ((net.actor, actor), (net.critic, critic)):
        for i, lin in enumerate([m for m in ref if isinstance(m, torch.nn.Linear)]):
            np.testing.assert_allclose(f.weight(i).cpu().numpy(), lin.weight.detach().numpy(), rtol=5e-5, atol=5e-6)
            np.testing.assert_allclose(f.bias(i).cpu().numpy(), lin.bias.detach().numpy(), rtol=5e-5, atol=5e-6)
    with pytest.raises(ValueError):
        with policy_within_training_step(algo):
            algo.update(buf, batch_size + 1, 1)


@pytest.mark.parametrize("D,A,M,variant", [(48, 5, 1000, "default"), (18, 3, 32, "dual"), (64, 16, 4103, "nonorm"),
                                           (5, 4, 70, "pg"), (48, 5, 65536, "default")])
def test_actor_rows_kernel_matches_float64_autograd(D, A, M, variant):
    """csrc/ppo_rows.hip: forward + policy loss + backward of a D-128-128-A actor in one launch vs float64 autograd of
    the reference's arithmetic (ppo.py:183-196, 210): gradient slabs (summed), clip objective and entropy sums."""
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.utils.net import FlatMLP

    H = 128
    rng = np.random.default_rng(D + A + M)
    f = FlatMLP([D, H, H, A], device=DEV, seed=3)
    n = M + 37
    obs = rng.standard_normal((n, D)).astype(np.float32)
    act = rng.integers(0, A, n)
    logp_old = (rng.standard_normal(n) * 0.3 - 1.3).astype(np.float32)
    adv = (rng.standard_normal(n) * 2 + 0.3).astype(np.float32)
    perm = rng.permutation(n)[:M]
    kw = dict(default={}, dual=dict(dual_clip=2.0, eps_clip=0.1), nonorm=dict(adv_norm=False, ent_coef=0.03),
              pg=dict(loss_kind=1, adv_norm=False))[variant]
    cfg = ops.make_ppo_cfg(**kw)
    d = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x)).to(DEV, dt)  # noqa: E731
    adv_d, perm_d = d(adv), d(perm)
    stats = ops.ppo_adv_stats(adv_d, d(np.array([0, M], np.int64)), perm=perm_d, max_rows=M) if kw.get("adv_norm", True) else None
    slabs, partial = ops.ppo_actor_rows_update(f.flat.data, d(obs), d(act, torch.int32), d(logp_old), adv_d, cfg, A, H,
                                               adv_stats=None if stats is None else stats[0], perm=perm_d)
    n_cu = ops.device_info()["n_cu"]
    assert slabs.shape[0] == ops.ppo_actor_rows_grid(M) and slabs.shape[0] in (min(-(-M // 32), n_cu), min(-(-M // 64), n_cu))
    # float64 replica
    lins = []
    for i in range(3):
        lin = torch.nn.Linear(f.dims[i], f.dims[i + 1]).double()
        with torch.no_grad():
            lin.weight.copy_(f.weight(i).cpu().double())
            lin.bias.copy_(f.bias(i).cpu().double())
        lins.append(lin)
    x = torch.as_tensor(obs[perm]).double()
    lsm = torch.log_softmax(lins[2](torch.relu(lins[1](torch.relu(lins[0](x))))), -1)
    logp = lsm.gather(1, torch.as_tensor(act[perm]).view(-1, 1)).flatten()
    a = torch.as_tensor(adv[perm]).double()
    if kw.get("adv_norm", True):
        a = (a - a.mean()) / (a.std() + 1e-8)
    ent = -(lsm.exp() * lsm).sum(-1)
    if variant == "pg":
        obj = logp * a
    else:
        eps = kw.get("eps_clip", 0.2)
        ratio = (logp - torch.as_tensor(logp_old[perm]).double()).exp()
        obj = torch.min(ratio * a, ratio.clamp(1 - eps, 1 + eps) * a)
        if kw.get("dual_clip"):
            obj = torch.where(a < 0, torch.max(obj, kw["dual_clip"] * a), obj)
    loss = -obj.mean() - kw.get("ent_coef", 0.01) * ent.mean()
    loss.backward()
    g_ref = torch.cat([t.grad.flatten() for lin in lins for t in (lin.weight, lin.bias)]).numpy()
    g = slabs.double().sum(0).cpu().numpy()
    assert np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref) < 2e-5
    np.testing.assert_allclose(g, g_ref, rtol=1e-3, atol=2e-4 * np.abs(g_ref).max())
    p = partial.view(-1, 4).sum(0).cpu().numpy()
    np.testing.assert_allclose([p[0], p[2]], [float(obj.sum()), float(ent.sum())], rtol=2e-5)
    assert p[1] == 0 and p[3] == 0
    # deterministic: the same launch twice gives the same bits
    slabs2, _ = ops.ppo_actor_rows_update(f.flat.data, d(obs), d(act, torch.int32), d(logp_old), adv_d, cfg, A, H,
                                          adv_stats=None if stats is None else stats[0], perm=perm_d)
    assert torch.equal(slabs, slabs2)


@pytest.mark.parametrize("gen", [1, 2, 3])
@pytest.mark.parametrize("D,N,Mr,vclip", [(48, 8, 1000, True), (18, 1, 300, False), (18, 3, 77, True), (48, 8, 8192, False),
                                          (6, 2, 32, False), (20, 4, 2100, True)])
def test_critic_rows_kernel_matches_float64_autograd(D, N, Mr, vclip, gen):
    """Critic: value = MLP(joint row), value term (with / without clipping) for the row's N agents and the critic's backward
    pass vs float64 autograd.  gen 1: csrc/ppo_rows.hip, one launch, layer-1 weights streamed in 32-column slices, a full
    gradient slab per workgroup; gen 2: csrc/critic_train.hip + critic_dw1.hip (weights in registers, dW1 as a split-K
    pass over the published dH1): two slab arrays, W1 | rest; gen 3 (round 4, opt-in: less traffic, 2 us slower): dW2 in the split-K pass too
    (published H1 / dH2), three slab arrays."""
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.utils.net import FlatMLP

    H, K1 = 128, N * D
    if gen == 1 and not ops.ppo_critic_rows_supported(K1, [H, H], N, "relu"):
        pytest.skip("the one-launch critic step of round 2 serves input widths up to 96 (wider ones: gen 2)")
    rng = np.random.default_rng(D + N + Mr)
    f = FlatMLP([K1, H, H, 1], device=DEV, seed=4)
    n_rows = Mr + 13
    obs = rng.standard_normal((n_rows, K1)).astype(np.float32)
    ret = (rng.standard_normal(n_rows * N) * 2).astype(np.float32)
    v_old = rng.standard_normal(n_rows * N).astype(np.float32)
    rows = rng.permutation(n_rows)[:Mr]
    cfg = ops.make_ppo_cfg(value_clip=vclip, vf_coef=0.7, eps_clip=0.3)
    d = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x)).to(DEV, dt)  # noqa: E731
    def run():
        if gen == 1:
            sl, part = ops.ppo_critic_rows_update(f.flat.data, d(obs), d(ret), cfg, N, H, v_s_old=d(v_old) if vclip else None,
                                                  rows=d(rows))
            return sl.double().sum(0), part, sl
        ws: dict = {}
        w1, rest, part = ops.critic_rows_grad_ppo(f.flat.data, d(obs), d(ret), cfg, N, H, v_s_old=d(v_old) if vclip else None,
                                                  rows=d(rows), ws=ws, split_dw2=gen == 3)
        assert w1.shape[1] == H * K1 and w1.shape[1] + rest.shape[1] == f.flat.numel()
        g_rest, extra = rest.double().sum(0), []
        if gen == 3:  # gen 3: dW2 from the split-K pass (published H1 / dH2); the W2 columns of the rest slabs are never written
            w2 = next(iter(ws.values()))["w2"]
            assert torch.count_nonzero(rest[:, H:H + H * H]) == 0 and w2.shape[1] == H * H
            g_rest[H:H + H * H] += w2.double().sum(0)
            extra = [w2.reshape(-1)]
            # the optimizer's segments for this mode tile the critic's parameters exactly once
            segs = ops.critic_grad_segs(next(iter(ws.values())), 0, K1, H, 1)
            assert [sg[1] for sg in segs] == [0, H * K1, H * K1 + H, H * K1 + H + H * H] and sum(sg[2] for sg in segs) == f.flat.numel()
            flat_g = ops.reduce_slabs_segs(segs, f.flat.numel())
            assert torch.allclose(flat_g.double(), torch.cat([w1.double().sum(0), g_rest]), rtol=1e-4, atol=1e-7)
        return torch.cat([w1.double().sum(0), g_rest]), part, torch.cat([w1.reshape(-1), rest.reshape(-1), *extra]).clone()

    g_sum, partial, slabs = run()
    lins = []
    for i in range(3):
        lin = torch.nn.Linear(f.dims[i], f.dims[i + 1]).double()
        with torch.no_grad():
            lin.weight.copy_(f.weight(i).cpu().double())
            lin.bias.copy_(f.bias(i).cpu().double())
        lins.append(lin)
    x = torch.as_tensor(obs[rows]).double()
    v = lins[2](torch.relu(lins[1](torch.relu(lins[0](x))))).expand(Mr, N)
    sid = rows[:, None] * N + np.arange(N)[None, :]
    r_t, vo = torch.as_tensor(ret[sid]).double(), torch.as_tensor(v_old[sid]).double()
    if vclip:
        v_clip = vo + (v - vo).clamp(-0.3, 0.3)
        vf = torch.max((r_t - v) ** 2, (r_t - v_clip) ** 2)
    else:
        vf = (r_t - v) ** 2
    (0.7 * vf.mean()).backward()
    g_ref = torch.cat([t.grad.flatten() for lin in lins for t in (lin.weight, lin.bias)]).numpy()
    g = g_sum.cpu().numpy()
    assert np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref) < 2e-5
    np.testing.assert_allclose(g, g_ref, rtol=1e-3, atol=2e-4 * np.abs(g_ref).max())
    p = partial.view(-1, 4).sum(0).cpu().numpy()
    np.testing.assert_allclose(p[1], float(vf.sum()), rtol=2e-5)
    assert p[0] == 0 and p[2] == 0
    assert torch.equal(slabs, run()[2])  # deterministic


@pytest.mark.parametrize("K1,Mr,n_a,ns_a", [(384, 8192, 23429, 256), (48, 300, 7000, 37), (384, 70000, 130, 3)])
def test_side_reductions_of_the_dw1_launch_give_the_optimizers_own_sums(K1, Mr, n_a, ns_a):
    """Extra workgroups of the dW1 launch sum other kernels' complete gradient slabs -- an actor's, and this step's own
    small-gradient slabs -- to ONE row each while the main workgroups wait on their loads (csrc/critic_dw1.hip), so that the
    optimizer launch reads a few MB instead of every slab.  Same additions in the same order as the optimizer's own sum: the
    rows equal `ops.reduce_slabs` bit for bit, and an Adam step over the one-row segments equals the step over the slabs, bit
    for bit (parameters and both moments), with and without a gradient-norm clip; the dW1 slabs themselves are untouched."""
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.utils.net import FlatMLP

    H = 128
    rng = np.random.default_rng(K1 + Mr)
    f = FlatMLP([K1, H, H, 1], device=DEV, seed=6)
    d = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x)).to(DEV, dt)  # noqa: E731
    obs, ret = d(rng.standard_normal((Mr + 5, K1)).astype(np.float32)), d(rng.standard_normal(Mr + 5).astype(np.float32))
    rows = d(rng.permutation(Mr + 5)[:Mr])
    cfg = ops.make_ppo_cfg()
    slabs_a = d((rng.standard_normal((ns_a, n_a + 3)) * 0.1).astype(np.float32))  # (a pitch wider than the segment)
    w1_plain, rest_plain, _ = [t.clone() for t in ops.critic_rows_grad_ppo(f.flat.data, obs, ret, cfg, 1, H, rows=rows)]
    n_rest = rest_plain.shape[1]
    red_a, red_c = torch.zeros(n_a, device=DEV), torch.zeros(n_rest, device=DEV)
    w1, rest, _ = ops.critic_rows_grad_ppo(f.flat.data, obs, ret, cfg, 1, H, rows=rows,
                                           side_reduce=[(slabs_a, red_a), ("rest", red_c)])
    assert torch.equal(w1, w1_plain) and torch.equal(rest, rest_plain)
    assert torch.equal(red_a, ops.reduce_slabs(slabs_a[:, :n_a].contiguous()))
    assert torch.equal(red_c, ops.reduce_slabs(rest_plain))
    nW1 = H * K1
    n = n_a + nW1 + n_rest
    for clip in (None, 0.7):
        outs = []
        for one_row in (False, True):
            p, m, v = torch.linspace(-1, 1, n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
            segs = [(red_a.view(1, -1), 0, n_a), (w1, n_a, nW1), (red_c.view(1, -1), n_a + nW1, n_rest)] if one_row else \
                [(slabs_a, 0, n_a), (w1, n_a, nW1), (rest, n_a + nW1, n_rest)]
            for step in range(2):
                ops.adam_step_segs(p, segs, m, v, step + 1, lr=1e-2, max_grad_norm=clip)
            outs.append((p, m, v))
        assert all(torch.equal(x, y) for x, y in zip(*outs)), clip


@pytest.mark.parametrize("glob,hidden,opts", [(False, (128, 128), {}), (True, (128, 128), dict(value_clip=True, max_grad_norm=0.5)),
                                              (False, (32, 32), dict(advantage_normalization=False))])
def test_generic_learn_as_one_graph_replay_equals_eager_launches(glob, hidden, opts):
    """`GenericPPO.learn(agent_batch)` (what the MARL trainers call, training_coordinator.py:336) from static buffers: the
    first call of a shape runs the launch sequence eagerly, the second captures it, later ones replay ONE hipGraph -- the same
    launches in the same order as `learn_steps`, so parameters, optimizer state and the returned losses are bit-identical to
    a twin that stays on eager launches, over four calls with changing rows (row kernels, dense path, centralized critic)."""
    N, Dd, n = 3, 18, 700
    outs = []
    for graph in (True, False):
        net = MLPActorCritic(Dd, 5, hidden, critic_obs_dim=N * Dd if glob else None, device=DEV, seed=3)
        algo = GenericPPO(net=net, critic_input="global" if glob else "local", n_agent=N if glob else 1, shuffle="device", seed=4,
                          graph=graph, lr=1e-3, **opts)
        gen = torch.Generator(device=DEV).manual_seed(9)
        losses = []
        for it in range(4):
            r = lambda *sh: torch.randn(*sh, device=DEV, generator=gen)  # noqa: E731
            b = Batch(obs=r(n, Dd), act=torch.randint(0, 5, (n,), device=DEV, generator=gen), rew=r(n), obs_next=r(n, Dd),
                      terminated=torch.rand(n, device=DEV, generator=gen) < 0.02, truncated=torch.zeros(n, dtype=torch.bool, device=DEV))
            if glob:
                b.global_obs, b.global_obs_next = r(n, N * Dd), r(n, N * Dd)
            losses.append(dict(algo.learn(b, 256, 2)))
        if graph:
            w = next(v for k, v in algo._ws.items() if isinstance(k, tuple) and k and k[0] == "glearn_graph")
            assert w["warm"] and "graph" in w
        outs.append((net.flat.data.clone(), algo.exp_avg_sq.clone(), algo.opt_step, losses, int(algo._perm_ctr.item())))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] == 4 * 2 * 2 and a[3] == b[3] and a[4] == b[4] == 8


@pytest.mark.parametrize("K1,Mr", [(384, 8192), (384, 77), (96, 1000), (72, 300), (48, 64), (18, 50)])
def test_critic_step_from_the_fragment_image_is_bit_identical_and_adam_keeps_the_image(K1, Mr):
    """The critic gradient step may take its first-layer weights from a FRAGMENT-ORDER copy (`ops.critic_w1_image`: coalesced
    16-B loads instead of a gather through sixteen 1.5 KB rows per 16-lane group; include/tsmarl.h).  Same values, same
    arithmetic: every output of the step is bit-identical to the gather's; and the segmented Adam launch that consumes the
    step's slabs leaves the image equal to one rebuilt from the updated flat vector (with and without a gradient-norm clip),
    three steps in a row, for widths with and without zero-padded k-groups."""
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.utils.net import FlatMLP

    H, N = 128, 1
    rng = np.random.default_rng(K1 + Mr)
    f = FlatMLP([K1, H, H, 1], device=DEV, seed=5)
    d = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x)).to(DEV, dt)  # noqa: E731
    obs, ret = d(rng.standard_normal((Mr + 9, K1)).astype(np.float32)), d(rng.standard_normal(Mr + 9).astype(np.float32))
    rows = d(rng.permutation(Mr + 9)[:Mr])
    cfg = ops.make_ppo_cfg()
    img = ops.critic_w1_image(f.flat.data, K1)
    kj = ops.call("tsm_critic_rows_w1_image_kj", K1)
    assert img.numel() == H * 16 * kj
    # the layout the header documents: element (w, j, lane, i) = w0[16 w + lane % 16][16 j + 4 (lane / 16) + i], zero pads
    w0 = f.flat.data[:H * K1].view(H, K1).cpu().numpy()
    pad = np.zeros((H, 16 * kj), np.float32)
    pad[:, :K1] = w0
    want = pad.reshape(8, 16, kj, 4, 4).transpose(0, 2, 3, 1, 4)  # [w][o16][j][kq][i] -> [w][j][kq][o16][i]
    assert np.array_equal(img.cpu().numpy().reshape(8, kj, 4, 16, 4), want)
    nW1 = H * K1
    for clip in (None, 0.5):
        p = f.flat.data.clone()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        im = ops.critic_w1_image(p, K1)
        for step in range(3):
            a = [t.clone() for t in ops.critic_rows_grad_ppo(p, obs, ret, cfg, N, H, rows=rows)]
            ws: dict = {}
            b = ops.critic_rows_grad_ppo(p, obs, ret, cfg, N, H, rows=rows, ws=ws, w1_image=im)
            wsb = next(iter(ws.values()))
            assert all(torch.equal(x, y) for x, y in zip(a, b))
            ops.adam_step_segs(p, [(b[0], 0, nW1, None, im), (b[1], nW1, p.numel() - nW1)], m, v, step + 1, lr=1e-2,
                               max_grad_norm=clip)
            assert torch.equal(im, ops.critic_w1_image(p, K1)), (clip, step)
            assert wsb["dh1"].shape == (Mr, H)


@pytest.mark.parametrize("glob,hidden,max_cycles,T,graph", [
    (True, (128, 128), 10, 10, True),    # aligned: episodes end at the last slot only -> the full pass is skipped
    (True, (128, 128), 7, 10, True),     # an episode ends mid-buffer -> device flag -> the full pass runs after all
    (False, (64, 64), 10, 10, False), (False, (64, 64), 4, 10, True)])
def test_next_values_from_the_next_slot_equal_the_full_critic_pass(glob, hidden, max_cycles, T, graph):
    """Rows written by a Collector are chained (obs_next of slot t is obs of slot t + 1 unless the episode ended), so
    GenericPPO takes V(obs_next) from V(obs) of the next slot + a pass over the last slot (ops.value_next_select) instead
    of a second pass over every row (a2c.py:123-124).  The update must be bit-identical to the one that runs both passes,
    over three collect + update rounds (graph replay included), aligned or not."""
    n_env, N = 48, 3
    finals = []
    for shift in (True, False):
        env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=max_cycles, device=DEV, seed=4)
        D = env.obs_dim
        net = MLPActorCritic(D, 5, hidden, critic_obs_dim=N * D if glob else None, device=DEV, seed=1)
        algo = GenericPPO(net=net, critic_input="global" if glob else "local", n_agent=N, shuffle="device", seed=2,
                          dispatch="pooled" if glob else "per_agent", graph=graph, shift_next_values=shift)
        buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, D, device=DEV)
        col = Collector(algo, env, buf)
        col.reset()
        outs = []
        for _ in range(3):
            with policy_within_training_step(algo):
                col.collect(n_step=n_env * T)
                assert buf.rows_chained is True
                if shift:  # the two ways to V(obs_next) agree on this very batch
                    pb = algo._preprocess_batch(buf, uniform_T=T)
                    algo.shift_next_values = False
                    ref = algo._preprocess_batch(buf, uniform_T=T)
                    algo.shift_next_values = True
                    assert torch.equal(pb["ret"], ref["ret"]) and torch.equal(pb["adv"], ref["adv"])
                ts = algo.update(buf, batch_size=480, repeat=1)
            col.reset_buffer(keep_statistics=True)
            outs.append(ts.get_loss_stats_dict())
        finals.append((net.flat.data.clone(), outs))
    assert torch.equal(finals[0][0], finals[1][0])
    assert finals[0][1] == finals[1][1]


def test_rows_added_outside_a_collector_are_not_treated_as_chained():
    n_env, N, T = 8, 3, 5
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=4)
    buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, env.obs_dim, device=DEV)
    assert buf.rows_chained == "empty"
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)  # noqa: E731
    buf.add_device(z(n_env, N, env.obs_dim), z(n_env, N, dt=torch.int32), z(n_env, N), z(n_env, N, dt=torch.uint8),
                   z(n_env, N, dt=torch.uint8), z(n_env, N, env.obs_dim))
    assert buf.rows_chained is False  # arbitrary rows: obs_next need not be the next slot's obs
    buf.reset()
    assert buf.rows_chained == "empty"
    net = MLPActorCritic(env.obs_dim, 5, (64, 64), device=DEV, seed=1)
    algo = GenericPPO(net=net, seed=2)
    col = Collector(algo, env, buf)
    col.reset()
    col.collect(n_step=n_env * 2)
    assert buf.rows_chained is True
    col.reset_env()                  # the next rows do not continue the stored ones
    col.collect(n_step=n_env * 2)
    assert buf.rows_chained is False


def test_graph_captured_over_chained_rows_is_not_replayed_on_unchained_rows():
    """The update graph bakes in, at capture time, whether V(obs_next) comes from the next slot's V(obs)
    (`buffer.rows_chained`).  Two half-collects with `reset_env()` between them fill the same buffer shape with rows that do
    NOT continue one another: the chained graph must not be replayed on them (its key carries the marker).  The update
    has to equal the one of an algorithm that always runs both critic passes (shift_next_values=False)."""
    n_env, N, T = 48, 3, 10
    finals = []
    for shift in (True, False):
        env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=25, device=DEV, seed=4)
        D = env.obs_dim
        net = MLPActorCritic(D, 5, (128, 128), critic_obs_dim=N * D, device=DEV, seed=1)
        algo = GenericPPO(net=net, critic_input="global", n_agent=N, shuffle="device", seed=2, dispatch="pooled", graph=True,
                          shift_next_values=shift)
        buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, D, device=DEV)
        col = Collector(algo, env, buf)
        col.reset()
        outs = []
        for it in range(5):
            with policy_within_training_step(algo):
                if it < 3:  # eager, capture, replay: on chained rows
                    col.collect(n_step=n_env * T)
                    assert buf.rows_chained is True
                else:       # the same shape, but slot T/2 does not continue slot T/2 - 1
                    col.collect(n_step=n_env * (T // 2))
                    col.reset_env()
                    col.collect(n_step=n_env * (T - T // 2))
                    assert buf.rows_chained is False
                ts = algo.update(buf, batch_size=480, repeat=1)
            col.reset_buffer(keep_statistics=True)
            outs.append(ts.get_loss_stats_dict())
        finals.append((net.flat.data.clone(), outs))
    assert torch.equal(finals[0][0], finals[1][0])
    assert finals[0][1] == finals[1][1]


@pytest.mark.parametrize("K1,Mr,perm", [(384, 4099, False), (384, 70000, True), (48, 1000, True), (18, 300, False), (20, 31, True),
                                        (6, 1, False), (200, 257, True)])
def test_critic_rows_forward_matches_float64(K1, Mr, perm):
    """csrc/critic_rows.hip: V(row) of an in-128-128-1 critic for many rows in one launch (layer-1 weights in registers,
    observation tile swizzled in LDS, layer 3 folded into layer 2's epilogue) vs the same MLP in float64 (a2c.py:121-127),
    1e-5 of the value scale; the dense GEMM chain it replaces agrees to f32 rounding; `run_if` = 0 leaves the output alone;
    misaligned parameter storage (the critic half of a joint flat vector starts at an odd offset) gives the same bits."""
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.utils.net import FlatMLP

    rng = np.random.default_rng(K1 + Mr)
    f = FlatMLP([K1, 128, 128, 1], device=DEV, seed=4)
    with torch.no_grad():
        for i in range(3):
            f.bias(i).copy_(torch.randn(f.bias(i).shape) * 0.1)
    n_rows = Mr + 5
    obs = torch.from_numpy(rng.standard_normal((n_rows, K1)).astype(np.float32)).to(DEV)
    rows = torch.from_numpy(rng.permutation(n_rows)[:Mr]).to(DEV) if perm else None
    v = ops.critic_rows_forward(f.flat.data, obs, 128, rows=rows, first_row=0 if perm else 3, Mr=Mr)
    x = (obs[rows] if perm else obs[3:3 + Mr]).double().cpu()
    W = [f.weight(i).double().cpu() for i in range(3)]
    b = [f.bias(i).double().cpu() for i in range(3)]
    ref = (torch.relu(torch.relu(x @ W[0].T + b[0]) @ W[1].T + b[1]) @ W[2].T + b[2]).reshape(-1)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(v.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * scale)
    dense = FlatMLP.forward(f, obs[rows] if perm else obs[3:3 + Mr].contiguous(), save=False).reshape(-1)
    np.testing.assert_allclose(v.cpu().numpy(), dense.cpu().numpy(), rtol=1e-5, atol=1e-5 * scale)
    # deterministic, and independent of how the rows fall into tiles / workgroups
    assert torch.equal(v, ops.critic_rows_forward(f.flat.data, obs, 128, rows=rows, first_row=0 if perm else 3, Mr=Mr))
    if Mr > 40:
        part = ops.critic_rows_forward(f.flat.data, obs, 128, rows=None if rows is None else rows[7:].contiguous(),
                                       first_row=0 if perm else 10, Mr=Mr - 7)
        assert torch.equal(part, v[7:])
    # a pass a captured graph carries for the rows that need it: flag 0 = no-op
    out = torch.full((Mr,), -7.0, device=DEV)
    ops.critic_rows_forward(f.flat.data, obs, 128, rows=rows, first_row=0 if perm else 3, Mr=Mr,
                            run_if=torch.zeros(1, dtype=torch.int32, device=DEV), out=out)
    assert bool((out == -7.0).all())
    ops.critic_rows_forward(f.flat.data, obs, 128, rows=rows, first_row=0 if perm else 3, Mr=Mr,
                            run_if=torch.ones(1, dtype=torch.int32, device=DEV), out=out)
    assert torch.equal(out, v)
    # parameters at a 4-byte (not 16-byte) aligned address
    pad = torch.zeros(f.flat.numel() + 1, device=DEV)
    pad[1:].copy_(f.flat.data)
    assert torch.equal(ops.critic_rows_forward(pad[1:], obs, 128, rows=rows, first_row=0 if perm else 3, Mr=Mr), v)
    with pytest.raises(ValueError):
        ops.critic_rows_forward(f.flat.data[:-1], obs, 128)


@pytest.mark.parametrize("K1,n_out,T,E,N", [(384, 8, 25, 40, 8), (48, 3, 7, 9, 3), (18, 1, 5, 13, 1), (384, 8, 1, 70, 8), (20, 2, 33, 3, 2),
                                            (384, 8, 25, 400, 8)])
def test_ctde_critic_rows_kernel_matches_float64_autograd(K1, n_out, T, E, N):
    """csrc/critic_train.hip (LOSS 1) + critic_dw1.hip: the critic half of CTDEPolicy.learn (ctde.py:149-172) on chained rows in one launch --
    values = critic(global_obs).mean(1), the TD target from the NEXT row's value (same forward pass, halo row per tile;
    v_last for the last slot), MSE, backward -- vs float64 autograd of the same arithmetic."""
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.utils.net import FlatMLP

    rng = np.random.default_rng(K1 + T + E)
    f = FlatMLP([K1, 128, 128, n_out], device=DEV, seed=4)
    with torch.no_grad():
        for i in range(3):
            f.bias(i).copy_(torch.randn(f.bias(i).shape) * 0.1)
    joint = rng.standard_normal((T, E, K1)).astype(np.float32)
    # keep every pre-activation away from the ReLU kink: within ~1e-7 of zero the f32 kernel and the f64 reference take
    # different sides and a whole unit's gradient row differs (seen at 10 000 rows x 256 units: |z2| = 3e-8)
    W = [f.weight(i).double().cpu() for i in range(2)]
    bb = [f.bias(i).double().cpu() for i in range(2)]
    for _ in range(8):
        z1 = torch.as_tensor(joint.reshape(-1, K1)).double() @ W[0].T + bb[0]
        z2 = torch.relu(z1) @ W[1].T + bb[1]
        near = ((z1.abs() < 1e-5).any(1) | (z2.abs() < 1e-5).any(1)).numpy()
        if not near.any():
            break
        joint.reshape(-1, K1)[near] = rng.standard_normal((int(near.sum()), K1)).astype(np.float32)
    assert not near.any()
    rew = rng.standard_normal((T, E, N)).astype(np.float32)
    term = (rng.random((T, E, N)) < 0.1)
    v_last = rng.standard_normal(E).astype(np.float32)
    a, gamma = N - 1, 0.97
    d = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x)).to(DEV, dt)  # noqa: E731
    run = lambda: ops.critic_rows_grad_td(f.flat.data, d(joint), T, E, d(rew), d(term.astype(np.uint8)), a, N, d(v_last),  # noqa: E731
                                          gamma, n_out)
    w1, rest, partial = run()
    keep = torch.cat([w1.reshape(-1), rest.reshape(-1)]).clone()
    lins = []
    for i in range(3):
        lin = torch.nn.Linear(f.dims[i], f.dims[i + 1]).double()
        with torch.no_grad():
            lin.weight.copy_(f.weight(i).cpu().double())
            lin.bias.copy_(f.bias(i).cpu().double())
        lins.append(lin)
    x = torch.as_tensor(joint).double().transpose(0, 1).reshape(E * T, K1)       # env-major rows (e, t)
    v = lins[2](torch.relu(lins[1](torch.relu(lins[0](x))))).mean(1).view(E, T)
    vn = torch.cat([v[:, 1:], torch.as_tensor(v_last).double().view(E, 1)], dim=1).detach()
    r_t = torch.as_tensor(rew[:, :, a]).double().t()
    nt = 1.0 - torch.as_tensor(term[:, :, a].astype(np.float64)).t()
    td = r_t + gamma * vn * nt
    loss = ((v - td) ** 2).mean()
    loss.backward()
    g_ref = torch.cat([t.grad.flatten() for lin in lins for t in (lin.weight, lin.bias)]).numpy()
    g = torch.cat([w1.double().sum(0), rest.double().sum(0)]).cpu().numpy()
    assert np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref) < 2e-5
    np.testing.assert_allclose(g, g_ref, rtol=1e-3, atol=2e-4 * np.abs(g_ref).max())
    p = partial.view(-1, 4).sum(0).cpu().numpy()
    np.testing.assert_allclose(p[0], float((td - v).sum()), rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(p[1], float(((v - td) ** 2).sum()), rtol=2e-5)
    w1b, restb, _ = run()
    assert torch.equal(keep, torch.cat([w1b.reshape(-1), restb.reshape(-1)]))
