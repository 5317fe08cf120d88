"""GPU tests (`-m gpu`) of the persistent rollout kernel (csrc/rollout.hip): one launch must reproduce,
bit for bit, the unfused sequence policy_forward -> mpe_step -> vrb_add of the Collector loop, and its extra
V(obs_next) output must equal a critic pass over the stored obs_next."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

DEV = "cuda"


def _job(n_env, N, T, fused, seed=3, slots=None, **ppo_kw):
    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=seed)
    net = DiscreteActorCritic(env.obs_dim, env.n_act, 64, device=DEV, seed=seed)
    algo = PPO(net=net, seed=seed, **ppo_kw)
    buf = DeviceVectorReplayBuffer(n_env * (slots or T), n_env, N, env.obs_dim, device=DEV)
    col = Collector(algo, env, buf, fused_rollout=fused, use_graph=False)
    col.reset()
    return env, net, algo, buf, col


# (1024, 3): 205 workgroups -> the eight-wave form of the kernel; (1400, 3): 280 workgroups of 82 KB LDS -> eight waves (70 KB and the
# four-wave form until round 5's pair-force scratch); (2100, 2): 263 workgroups of < 80 KB -> the four-wave form (two workgroups per
# CU: its env / uniform / payload lanes sit on waves 1 and 2); (600, 8): 300 workgroups of 91 KB LDS -> eight waves again
@pytest.mark.parametrize("n_env,N,T,steps", [(64, 3, 25, 25), (7, 3, 6, 15), (33, 8, 5, 12), (5, 1, 4, 9), (10, 2, 7, 7),
                                             (1024, 3, 25, 25), (1400, 3, 5, 7), (600, 8, 4, 6), (37, 6, 5, 7), (41, 5, 4, 6),
                                             (130, 4, 6, 8), (300, 3, 5, 6), (600, 3, 5, 6), (2100, 2, 4, 6)])
@pytest.mark.parametrize("form", ["wave", "tile"])
def test_fused_rollout_is_bit_identical_to_unfused(n_env, N, T, steps, form):
    """Both forms of the 64-wide persistent rollout (option "rollout_form": the wave-autonomous one -- a wave owns 16 // N whole envs
    and runs actor, critic, heads, env step and buffer rows by itself; workgroups of 1, 2 or 4 such waves: (300, 3) / (600, 3) /
    (1024, 3) -- and the eight- / four-wave tile form) against the unfused launch sequence."""
    slots = steps + 3 + 1  # both collects fit without wrap-around
    outs = []
    for fused in (False, True):
        env, net, algo, buf, col = _job(n_env, N, T, fused, slots=slots)
        assert col._can_fuse() == fused
        with policy_within_training_step(algo), ops.kernel_override(rollout_form=1 if form == "tile" else 2):
            st1 = col.collect(n_step=n_env * steps)
            st2 = col.collect(n_step=n_env * 3)  # continues mid-episode, crosses resets for short horizons
        outs.append(dict(
            obs=buf.obs_store.clone(), obs_next=buf.obs_next_store.clone(), act=buf.act_store.clone(),
            rew=buf.rew_store.clone(), trunc=buf.trunc_store.clone(), term=buf.term_store.clone(),
            logp=buf.logp_store.clone(), vs=buf.vs_store.clone(), done=buf.done_store.clone(),
            state=buf.index.state.clone(), apos=env.agent_pos.clone(), avel=env.agent_vel.clone(),
            lpos=env.landmark_pos.clone(), steps=env.steps.clone(), ep=env.episode_ctr.clone(), obs_cur=env.obs_cur.clone(),
            tick=env.rng_tick.clone(), ret1=st1.returns, len1=st1.lens, n1=st1.n_collected_episodes,
            ret2=st2.returns, n2=st2.n_collected_episodes))
        if fused:
            vnext = buf.vnext_store.clone()
            # V(obs_next) == a critic pass over the stored obs_next rows (a2c.py:124)
            n_rows = steps + 3
            ref = ops.policy_forward(net.flat.data, buf.obs_next_store[:n_rows].reshape(-1, env.obs_dim), 5, 64,
                                     mode="none")["value"].reshape(n_rows, n_env, N)
            assert torch.equal(vnext[:n_rows], ref)
            assert buf.policy_outputs_version == algo.param_version
        else:
            assert buf.policy_outputs_version is None
    a, b = outs
    for k in a:
        if isinstance(a[k], torch.Tensor):
            assert torch.equal(a[k], b[k]), k
        elif isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


@pytest.mark.parametrize("path", ["fused", "graph", "eager"])
def test_lens_behind_a_statistics_keeping_buffer_reset_count_buffered_rows(path):
    """collector.py:203: `lens` = the episode's rows in the buffer.  Episodes of 7 steps, collects of 5 vector steps with
    `reset_buffer(keep_statistics=True)` in between (the trainer's sequence): the episodes that end in the second collect have 2 rows
    in the buffer and their WHOLE return -- on the persistent rollout (compact episode record), the captured three-launch loop and
    the eager one (dense per-step arrays) alike.  (The host path meets the reference's own run: test_gpu_marl.py.)"""
    env, net, algo, buf, col = _job(64, 3, 7, path == "fused", slots=8)
    col.use_graph = path == "graph"
    outs = []
    with policy_within_training_step(algo):
        for k in range(4):   # (the graph path captures on its second call of a shape)
            st = col.collect(n_step=64 * 5)
            outs.append((np.asarray(st.lens).copy(), np.asarray(st.returns).copy(), buf.rew_store[:5].clone()))
            col.reset_buffer(keep_statistics=True)
    assert len(outs[0][0]) == 0                                   # 5 steps: no episode has ended
    assert np.array_equal(outs[1][0], np.full(64, 2))             # ends at step 7 = row 2 of the second collect; 7 - 5 rows buffered
    assert np.array_equal(outs[2][0], np.full(64, 4))             # next episode: steps 8..14, rows 3..5 of collect 2 are gone: 4 rows
    # ... while the return is the whole episode's: rows 0..4 of the first collect + rows 0..1 of the second
    want = (outs[0][2].double().sum(0) + outs[1][2][:2].double().sum(0)).cpu().numpy()
    np.testing.assert_allclose(outs[1][1], want, rtol=1e-12)


def test_update_from_stored_rollout_outputs_equals_recomputed():
    """PPO.update fed by the rollout kernel's stored logp / v_s / V(obs_next) == update that recomputes them."""
    res = []
    for fused in (False, True):
        env, net, algo, buf, col = _job(48, 3, 25, fused, shuffle="numpy", dispatch="per_agent", use_graph=True)
        np.random.seed(1)
        for _ in range(3):
            with policy_within_training_step(algo):
                col.collect(n_step=48 * 25)
                st = algo.update(buf, 256, 2)
            col.reset_buffer(keep_statistics=True)
        res.append((net.flat.data.clone(), st.get_loss_stats_dict()))
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][1] == res[1][1]


# ---- the actor-only persistent rollout for 128-wide actors (csrc/rollout_rows.hip) ----
def _job128(n_env, N, T, fused, glob, seed=3, slots=None):
    from tianshou_marl_amd.algorithm import GenericPPO
    from tianshou_marl_amd.utils.net import MLPActorCritic

    env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=seed)
    net = MLPActorCritic(env.obs_dim, 5, (128, 128), critic_obs_dim=N * env.obs_dim if glob else None, device=DEV, seed=seed)
    algo = GenericPPO(net=net, critic_input="global" if glob else "local", n_agent=N, seed=seed, shuffle="numpy",
                      dispatch="pooled")
    buf = DeviceVectorReplayBuffer(n_env * (slots or T), n_env, N, env.obs_dim, device=DEV)
    col = Collector(algo, env, buf, fused_rollout=fused, use_graph=False)
    col.reset()
    return env, net, algo, buf, col


# (600, 8): 38 workgroups of 16 envs; (50, 3): 42 envs per workgroup, 126 live rows (partial last tile); (4096, 8): the
# BASELINE configs[2] size, one workgroup per CU
# (40, 5) / (21, 7): observation widths 30 / 42 -- a thread's observation elements straddle the 32-row tiles
@pytest.mark.parametrize("n_env,N,T,steps,glob", [(600, 8, 4, 6, True), (50, 3, 6, 15, False), (5, 1, 4, 9, False),
                                                  (33, 8, 25, 25, True), (4096, 8, 25, 25, True), (40, 5, 5, 8, False),
                                                  (21, 7, 4, 6, True), (37, 6, 5, 7, False), (9, 2, 3, 7, False),
                                                  (130, 4, 6, 8, True)])
@pytest.mark.parametrize("form", ["wave", "tile"])
def test_actor_rollout_is_bit_identical_to_unfused(n_env, N, T, steps, glob, form):
    """Both forms of the actor-only rollout (option "rollout_form": the wave-autonomous default -- a wave owns 16 // N whole
    envs, transposed products, no workgroup barrier in the step loop -- and round 2's tile form) against the unfused launch
    sequence.  (37, 6) / (9, 2) / (130, 4): 2 / 8 / 4 envs per wave with 12 / 16 / 16 live rows, partial last waves."""
    from tianshou_marl_amd import ops

    slots = steps + 3 + 1
    outs = []
    for fused in (False, True):
        env, net, algo, buf, col = _job128(n_env, N, T, fused, glob, slots=slots)
        assert col._can_fuse() == fused and col._can_fuse_actor() == fused
        with policy_within_training_step(algo), ops.kernel_override(rollout_form=1 if form == "tile" else 2):
            st1 = col.collect(n_step=n_env * steps)
            st2 = col.collect(n_step=n_env * 3)
        outs.append(dict(
            obs=buf.obs_store.clone(), obs_next=buf.obs_next_store.clone(), act=buf.act_store.clone(),
            rew=buf.rew_store.clone(), trunc=buf.trunc_store.clone(), term=buf.term_store.clone(),
            logp=buf.logp_store.clone(), done=buf.done_store.clone(), state=buf.index.state.clone(),
            apos=env.agent_pos.clone(), avel=env.agent_vel.clone(), lpos=env.landmark_pos.clone(), steps=env.steps.clone(),
            ep=env.episode_ctr.clone(), obs_cur=env.obs_cur.clone(), tick=env.rng_tick.clone(), ret1=st1.returns,
            len1=st1.lens, n1=st1.n_collected_episodes, ret2=st2.returns, n2=st2.n_collected_episodes))
        # logp is the acting policy's in both paths; only the unfused path also stored V(obs)
        assert buf.logp_outputs_version == algo.param_version
        assert buf.behaviour_outputs_version == (None if fused else algo.param_version)
    a, b = outs
    for k in a:
        if isinstance(a[k], torch.Tensor):
            assert torch.equal(a[k], b[k]), k
        elif isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


def test_update_after_actor_rollout_equals_update_after_unfused_collect():
    """The update recomputes V(obs) itself after the actor-only rollout and takes logp_old from the buffer: same weights
    as after the unfused collect (which stored both)."""
    res = []
    for fused in (False, True):
        env, net, algo, buf, col = _job128(64, 8, 25, fused, True)
        np.random.seed(2)
        for _ in range(3):
            with policy_within_training_step(algo):
                col.collect(n_step=64 * 25)
                st = algo.update(buf, 512 * 8, 2)
            col.reset_buffer(keep_statistics=True)
        res.append((net.flat.data.clone(), st.get_loss_stats_dict()))
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][1] == res[1][1]


@pytest.mark.parametrize("kind", ["shared_ppo64", "ctde", "shared_ctde"])
def test_wrapped_and_ctde_policies_take_the_persistent_rollouts(kind):
    """A multi-agent wrapper in parameter-sharing mode forwards all rows to ONE policy (marl.py:137-190), and CTDEPolicy's
    DecentralizedActor (ctde.py:346-396) is a 48-128-128-5 actor like GenericPPO's: both collect through the persistent
    kernels, bit-identical to their unfused launch sequences (incl. a policy whose sampling counter was advanced by host
    forward calls before)."""
    from tianshou_marl_amd.algorithm.multiagent import CentralizedCritic, CTDEPolicy, DecentralizedActor, FlexibleMultiAgentPolicyManager

    n_env, N, T, steps = 96, 8 if "ctde" in kind else 3, 5, 8
    outs = []
    for fused in (False, True):
        env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=3)
        D = env.obs_dim
        if kind == "shared_ppo64":
            base = PPO(net=DiscreteActorCritic(D, 5, 64, device=DEV, seed=3), seed=3)
            base._sample_ctr = 1234  # as after host-side forward() calls
            pol = FlexibleMultiAgentPolicyManager(base, env, mode="shared")
        else:
            base = CTDEPolicy(actor=DecentralizedActor(D, 5, 128, device=DEV, seed=1),
                              critic=CentralizedCritic(N * D, N, 128, device=DEV, seed=2), seed=5)
            pol = FlexibleMultiAgentPolicyManager(base, env, mode="shared") if kind == "shared_ctde" else base
        buf = DeviceVectorReplayBuffer(n_env * (steps + 4), n_env, N, D, device=DEV)
        col = Collector(pol, env, buf, fused_rollout=fused, use_graph=False)
        col.reset()
        assert col._can_fuse() == fused
        with policy_within_training_step(pol):
            st1 = col.collect(n_step=n_env * steps)
            st2 = col.collect(n_step=n_env * 3)
        outs.append(dict(obs=buf.obs_store.clone(), obs_next=buf.obs_next_store.clone(), act=buf.act_store.clone(),
                         rew=buf.rew_store.clone(), trunc=buf.trunc_store.clone(), logp=buf.logp_store.clone(),
                         done=buf.done_store.clone(), state=buf.index.state.clone(), apos=env.agent_pos.clone(),
                         ep=env.episode_ctr.clone(), obs_cur=env.obs_cur.clone(), tick=env.rng_tick.clone(),
                         ret1=st1.returns, ret2=st2.returns, n=st1.n_collected_episodes + st2.n_collected_episodes))
    a, b = outs
    for k in a:
        if isinstance(a[k], torch.Tensor):
            assert torch.equal(a[k], b[k]), k
        elif isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


@pytest.mark.parametrize("kind", ["ppo64", "generic128", "generic128_global"])
def test_ignore_obs_next_buffer_collects_and_updates_with_the_reference_semantics(kind):
    """`VectorReplayBuffer(..., ignore_obs_next=True)` (buffer_base.py:612-616): no obs_next store; a row's obs_next is READ as
    obs[next(index)], next(index) being the row itself at an episode end and at the newest row.  The persistent rollouts
    skip the obs_next rows (half of their HBM writes) and leave every other store bit-identical; the update takes
    V(obs_next) from V(obs) at next(index) (tsm_value_next_index) -- bit-identical to an update on a full buffer whose
    obs_next rows hold exactly those observations.  Episodes end in the middle of the collected slots (max_cycles 10, 25
    slots)."""
    from tianshou_marl_amd.algorithm import GenericPPO
    from tianshou_marl_amd.utils.net import MLPActorCritic

    n_env, N, T, cyc = 48, 3, 25, 10
    finals = []
    for ign in (True, False):
        env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=cyc, device=DEV, seed=5)
        D = env.obs_dim
        if kind == "ppo64":
            net = DiscreteActorCritic(D, 5, 64, device=DEV, seed=2)
            algo = PPO(net=net, seed=7, shuffle="device", dispatch="per_agent")
        else:
            glob = kind.endswith("global")
            net = MLPActorCritic(D, 5, (128, 128), critic_obs_dim=N * D if glob else None, device=DEV, seed=2)
            algo = GenericPPO(net=net, critic_input="global" if glob else "local", n_agent=N, seed=7, shuffle="device",
                              dispatch="pooled" if glob else "per_agent")
        buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, D, device=DEV, ignore_obs_next=ign)
        col = Collector(algo, env, buf)
        col.reset()
        assert col._can_fuse()
        stores = []
        for _ in range(3):  # eager update, captured update, replay
            with policy_within_training_step(algo):
                col.collect(n_step=n_env * T)
                assert (buf.obs_next_store is None) == ign
                if not ign:  # what an ignore_obs_next buffer hands out as obs_next, written into the full buffer
                    done = buf.done_store[:T].bool()
                    self_next = done.clone()
                    self_next[T - 1] = True
                    nxt = torch.where(self_next, torch.arange(T, device=DEV).view(T, 1), torch.arange(T, device=DEV).view(T, 1) + 1)
                    buf.obs_next_store[:T] = buf.obs_store[:T][nxt.clamp(max=T - 1), torch.arange(n_env, device=DEV).view(1, -1)]
                    buf.policy_outputs_version = None   # (the rollout's V(obs_next) belongs to the true next observation)
                    buf.behaviour_outputs_version = None
                    buf.rows_chained = False
                else:
                    idx = np.arange(n_env * T)  # the reference layout on read: obs_next = obs[next(index)]
                    got = buf[idx]
                    assert np.array_equal(got.obs_next, got.obs[buf.index.next(torch.as_tensor(idx, device=DEV)).cpu().numpy()])
                stores.append([x[:T].clone() for x in (buf.obs_store, buf.act_store, buf.rew_store, buf.term_store,
                                                       buf.trunc_store, buf.logp_store, buf.done_store)])
                algo.update(buf, 480, 1)
            col.reset_buffer(keep_statistics=True)
        finals.append((net.flat.data.clone(), stores))
    assert torch.equal(finals[0][0], finals[1][0])
    for a, b in zip(finals[0][1], finals[1][1]):
        assert all(torch.equal(x, y) for x, y in zip(a, b))


# ---- kernels compiled for BASELINE's dimensions against the generic forms (option "generic_kernels") ----
def _snap(buf, net, extra=()):
    out = {k: getattr(buf, k).clone() for k in ("obs_store", "act_store", "rew_store", "done_store") if getattr(buf, k) is not None}
    for k in ("obs_next_store", "logp_store", "vs_store", "vnext_store"):
        if getattr(buf, k, None) is not None:
            out[k] = getattr(buf, k).clone()
    out["state"] = buf.index.state.clone()
    for i, n in enumerate(net if isinstance(net, (list, tuple)) else [net]):
        out[f"flat{i}"] = n.flat.data.clone()
    out.update(extra)
    return out


@pytest.mark.parametrize("case", ["spread_1024x3", "spread_actor_4096x8", "tag_512_3v1"])
def test_generic_and_specialised_kernels_give_the_same_bits(case):
    """VERDICT r4 'What's weak' 1b: the instantiations the benchmark runs (`ppo_update_split_kernel<64,0,18>`,
    `rollout_kernel<64,512,18,3>`, `policy_forward_kernel<64,18>` at configs[1]; `rollout_wave_kernel<3,8>` at configs[2];
    `<64,0,16>` / `<64,16>` at configs[4]) are compile-time specialisations of generic kernels: collect + update under the default
    rule and under `generic_kernels=1` must leave the same buffer rows, the same loss statistics and the same parameters."""
    res = []
    for generic in (0, 1):
        with ops.kernel_override(generic_kernels=generic):
            if case == "spread_1024x3":
                env, net, algo, buf, col = _job(1024, 3, 25, True, use_graph=True, shuffle="device")
                nets = [net]
                with policy_within_training_step(algo):
                    col.collect(n_step=1024 * 25)
                    st = algo.update(buf, 4096, 1)
                stats = st.get_loss_stats_dict()
            elif case == "spread_actor_4096x8":
                env, net, algo, buf, col = _job128(4096, 8, 25, True, True)
                nets = [net]
                with policy_within_training_step(algo):
                    col.collect(n_step=4096 * 25)
                stats = {}
            else:
                from tianshou_marl_amd.algorithm.multiagent.flexible_policy import FlexibleMultiAgentPolicyManager
                from tianshou_marl_amd.algorithm.multiagent.training_coordinator import agent_batches_from_buffer
                from tianshou_marl_amd.env.mpe_tag import DeviceSimpleTagVectorEnv

                env = DeviceSimpleTagVectorEnv(512, 1, 3, 2, max_cycles=25, device=DEV, seed=11)
                mk = lambda s: PPO(net=DiscreteActorCritic(env.obs_dim, 5, 64, device=DEV, seed=s), seed=s, shuffle="device")  # noqa: E731
                pols = {"adversaries": mk(21), "good": mk(22)}
                mgr = FlexibleMultiAgentPolicyManager(pols, env, mode="grouped", agent_groups=env.agent_groups)
                buf = DeviceVectorReplayBuffer(512 * 25, 512, env.n_agent, env.obs_dim, device=DEV)
                col = Collector(mgr, env, buf, fused_rollout=True, use_graph=False)
                col.reset()
                nets = [p.net for p in pols.values()]
                with policy_within_training_step(mgr):
                    col.collect(n_step=512 * 25)
                    batch = agent_batches_from_buffer(buf, env.agents, global_state=False)
                    stats = {name: dict(pols["good" if name.startswith("agent") else "adversaries"].learn(batch[name], 4096, 1))
                             for name in ("adversary_0", "agent_0")}
            torch.cuda.synchronize()
            res.append((_snap(buf, nets), stats))
    (a, sa), (b, sb) = res
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert sa == sb
