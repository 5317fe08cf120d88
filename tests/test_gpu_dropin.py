"""GPU tests of the drop-in boundary: a reference-shaped script -- `Net` /
`DiscreteActor` / `DiscreteCritic` / `DiscreteActorPolicy` / `AdamOptimizerFactory` / `PPO(policy=, critic=, optim=)` /
`VectorReplayBuffer(total, num)` / `Collector` / `run_training(OnPolicyTrainerParams)` -- runs on the device engine;
checkpoints move between the reference and the engine in both directions (tests/golden/checkpoint.npz holds the
reference's own `state_dict()` after its `mb64` update); LR schedules are followed by the captured update graph."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from test_gpu_ppo_replay import _flat, _job

    from tianshou_marl_amd.algorithm import PPO, GenericPPO, policy_within_training_step
    from tianshou_marl_amd.algorithm.multiagent.ctde import CentralizedCritic, CTDEPolicy, DecentralizedActor
    from tianshou_marl_amd.algorithm.optim import AdamOptimizerFactory, LRSchedulerFactoryLinear
    from tianshou_marl_amd.data import Collector, VectorReplayBuffer
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.env.spaces import Discrete
    from tianshou_marl_amd.trainer import OnPolicyTrainerParams
    from tianshou_marl_amd.utils.net import DiscreteActorCritic
    from tianshou_marl_amd.utils.ref_nets import ActorCritic, DiscreteActor, DiscreteActorPolicy, DiscreteCritic, Net

DEV = "cuda"


def _reference_shaped_algorithm(obs_dim, hidden, lr=1e-3, sched=None, **kw):
    """The model block of the reference's PPO script (test/discrete/test_ppo_discrete.py:91-137), imports changed only."""
    torch.manual_seed(0)
    net_a = Net(state_shape=(obs_dim,), hidden_sizes=hidden)
    net_c = Net(state_shape=(obs_dim,), hidden_sizes=hidden)
    actor = DiscreteActor(preprocess_net=net_a, action_shape=5, softmax_output=False)
    critic = DiscreteCritic(preprocess_net=net_c)
    for m in ActorCritic(actor, critic).modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    optim = AdamOptimizerFactory(lr=lr)
    if sched is not None:
        optim.with_lr_scheduler_factory(sched)
    policy = DiscreteActorPolicy(actor=actor, action_space=Discrete(5), deterministic_eval=True)
    return PPO(policy=policy, critic=critic, optim=optim, **kw), actor, critic


def test_reference_shaped_script_trains_through_run_training():
    n_env, T = 64, 25
    sched = LRSchedulerFactoryLinear(max_epochs=3, epoch_num_steps=2 * n_env * T, collection_step_num_env_steps=n_env * T)
    algorithm, actor, critic = _reference_shaped_algorithm(18, [64, 64], lr=1e-3, sched=sched, max_grad_norm=0.5,
                                                           shuffle="device")
    assert type(algorithm) is PPO and isinstance(algorithm.net, DiscreteActorCritic) and algorithm.dispatch == "pooled"
    # the parameters are the modules' (copied once into the flat HBM vector)
    np.testing.assert_array_equal(algorithm.net.view("actor.w0").cpu().numpy(), actor.preprocess.model.model[0].weight.detach().numpy())
    np.testing.assert_array_equal(algorithm.net.view("critic.w2").cpu().numpy(), critic.last.model[0].weight.detach().numpy())
    train_envs = DeviceSimpleSpreadVectorEnv(n_env, 3, device=DEV, seed=1)
    buf = VectorReplayBuffer(n_env * T, len(train_envs))
    assert not buf.allocated and len(buf) == 0
    train_collector = Collector(algorithm, train_envs, buf)
    assert buf.allocated and (buf.n_agent, buf.obs_dim) == (3, 18)
    seen = []
    result = algorithm.run_training(OnPolicyTrainerParams(
        train_collector=train_collector, max_epochs=2, epoch_num_steps=2 * n_env * T,
        collection_step_num_env_steps=n_env * T, batch_size=256, update_step_num_repetitions=2,
        multi_agent_return_reduction=lambda r: r.mean(axis=1), train_fn=lambda epoch, step: seen.append((epoch, step)),
        verbose=False))
    assert result.update_step == 4 and result.train_step == 4 * n_env * T and result.train_episode == 4 * n_env
    assert seen == [(1, 0), (1, 1600), (2, 3200), (2, 4800)]
    assert len(buf) == 0  # reset_buffer(keep_statistics=True) after every update (trainer.py:1104)
    # LambdaLR: lr = base * (1 - k / (ceil(3200 / 1600) * 3)) after k updates, on the host mirror and in HBM
    assert algorithm.lr == pytest.approx(1e-3 * (1 - 4 / 6)) and algorithm._lr_dev.item() == pytest.approx(algorithm.lr)
    # the schedule did not force a re-capture: one graph for the four updates
    assert sum(1 for k in algorithm._ws if isinstance(k, tuple) and k and k[0] == "graph") == 1
    assert torch.isfinite(algorithm.net.flat.data).all()


def test_wider_reference_nets_select_the_general_kernels():
    algorithm, actor, critic = _reference_shaped_algorithm(48, [128, 128], lr=3e-4)
    assert isinstance(algorithm, GenericPPO) and algorithm.net.actor.dims == [48, 128, 128, 5]
    # what the reference objects say carries over as for the 64-wide class: the policy's greedy evaluation and an update
    # on the whole batch handed over (ppo.py:55-133)
    assert algorithm.deterministic_eval is True and algorithm.dispatch == "pooled"
    obs = torch.randn(64, 48, device=DEV)
    out = algorithm.act_device(obs)  # outside a training step: the arg-max action (reinforce.py:185-189)
    assert torch.equal(out["act"].long(), out["logits"].argmax(-1))
    with policy_within_training_step(algorithm):
        sampled = algorithm.act_device(obs)["act"]
    assert not torch.equal(sampled.long(), out["logits"].argmax(-1))
    keys = [k for k in algorithm.state_dict() if k != "_optimizers"]
    ref = ["policy.actor." + k for k in actor.state_dict()] + ["critic." + k for k in critic.state_dict()]
    assert keys == ref
    # actor and critic trunks of different shapes are refused (one flat vector, one set of hidden sizes)
    mismatched = DiscreteActorPolicy(actor=DiscreteActor(preprocess_net=Net(state_shape=(18,), hidden_sizes=[64, 32]),
                                                         action_shape=5), action_space=Discrete(5))
    with pytest.raises(ValueError):
        PPO(policy=mismatched, critic=DiscreteCritic(preprocess_net=Net(state_shape=(18,), hidden_sizes=[64, 64])),
            optim=AdamOptimizerFactory())


def test_lr_schedule_is_followed_by_graph_and_eager_updates_alike():
    outs = []
    for use_graph in (True, False):
        n_env, T = 32, 25
        sched = LRSchedulerFactoryLinear(max_epochs=1, epoch_num_steps=5 * n_env * T, collection_step_num_env_steps=n_env * T)
        algo, _, _ = _reference_shaped_algorithm(18, [64, 64], lr=2e-3, sched=sched, use_graph=use_graph, shuffle="numpy",
                                                 seed=3)
        env = DeviceSimpleSpreadVectorEnv(n_env, 3, device=DEV, seed=2)
        col = Collector(algo, env, VectorReplayBuffer(n_env * T, n_env))
        col.reset()
        np.random.seed(0)
        lrs = []
        for _ in range(4):
            with policy_within_training_step(algo):
                col.collect(n_step=n_env * T)
                algo.update(col.buffer, 512, 1)
            col.reset_buffer(keep_statistics=True)
            lrs.append(algo.lr)
        outs.append((algo.net.flat.data.clone(), lrs))
    np.testing.assert_allclose(outs[0][1], [2e-3 * (1 - k / 5) for k in (1, 2, 3, 4)], rtol=1e-12)
    assert outs[0][1] == outs[1][1] and torch.equal(outs[0][0], outs[1][0])
    # and the schedule matters: a constant rate gives other weights
    algo, _, _ = _reference_shaped_algorithm(18, [64, 64], lr=2e-3, shuffle="numpy", seed=3)
    env = DeviceSimpleSpreadVectorEnv(32, 3, device=DEV, seed=2)
    col = Collector(algo, env, VectorReplayBuffer(32 * 25, 32))
    col.reset()
    np.random.seed(0)
    for _ in range(4):
        with policy_within_training_step(algo):
            col.collect(n_step=32 * 25)
            algo.update(col.buffer, 512, 1)
        col.reset_buffer(keep_statistics=True)
    assert not torch.equal(algo.net.flat.data, outs[0][0])


def test_checkpoint_interchange_with_the_reference(golden_dir):
    """(f)3: after replaying the reference's `mb64` update, `state_dict()` has the reference's keys, shapes, values and
    a torch-Adam `_optimizers` entry; the reference's own checkpoint loads into a fresh engine object."""
    g = np.load(os.path.join(golden_dir, "ppo_update.npz"), allow_pickle=True)
    ck = np.load(os.path.join(golden_dir, "checkpoint.npz"), allow_pickle=True)
    algo, net, buf = _job(g, "mb64", use_graph=True)
    np.random.seed(11)
    with policy_within_training_step(algo):
        algo.update(buf, 64, 2)
    sd = algo.state_dict()
    keys = [k for k in sd if k != "_optimizers"]
    assert keys == list(ck["ppo_keys"])
    for k in keys:
        assert tuple(sd[k].shape) == ck["ppo/" + k].shape, k
        np.testing.assert_allclose(sd[k].cpu().numpy(), ck["ppo/" + k], rtol=1e-5, atol=5e-6, err_msg=k)
    opt = sd["_optimizers"]
    assert len(opt) == 1 and {"state", "param_groups"} <= set(opt[0])
    grp = opt[0]["param_groups"][0]
    assert set(ck["ppo_opt_param_group_keys"]) <= set(grp) and grp["params"] == list(ck["ppo_opt_params"])
    np.testing.assert_allclose([grp["lr"], *grp["betas"], grp["eps"], grp["weight_decay"]], ck["ppo_opt_hyper"])
    assert len(opt[0]["state"]) == int(ck["ppo_opt_n_state"])
    for i, st in opt[0]["state"].items():
        assert float(st["step"]) == float(ck[f"ppo_opt/{i}/step"]) == 6.0
        for f in ("exp_avg", "exp_avg_sq"):
            ref = ck[f"ppo_opt/{i}/{f}"]
            assert tuple(st[f].shape) == ref.shape
            np.testing.assert_allclose(st[f].cpu().numpy(), ref, rtol=2e-4, atol=1e-7 if f == "exp_avg" else 1e-10, err_msg=f"{i}/{f}")
    # torch's own Adam accepts the entry (it IS the torch layout)
    params = [torch.nn.Parameter(torch.zeros_like(sd[k], device="cpu")) for k in keys]
    t_opt = torch.optim.Adam(params, lr=1.0)
    t_opt.load_state_dict({"state": {i: {k: v.cpu() for k, v in st.items()} for i, st in opt[0]["state"].items()},
                           "param_groups": opt[0]["param_groups"]})
    assert t_opt.param_groups[0]["lr"] == pytest.approx(3e-4) and float(t_opt.state[params[3]]["step"]) == 6.0
    # reference checkpoint -> fresh engine object
    ref_sd = {k: torch.from_numpy(ck["ppo/" + k]) for k in keys}
    ref_sd["_optimizers"] = [{"state": {i: {"step": torch.tensor(float(ck[f"ppo_opt/{i}/step"])),
                                            "exp_avg": torch.from_numpy(ck[f"ppo_opt/{i}/exp_avg"]),
                                            "exp_avg_sq": torch.from_numpy(ck[f"ppo_opt/{i}/exp_avg_sq"])}
                                        for i in range(int(ck["ppo_opt_n_state"]))},
                              "param_groups": [dict(lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                                                    params=list(range(12)))]}]
    fresh = PPO(net=DiscreteActorCritic(18, 5, 64, device=DEV, seed=9), lr=1.0)
    fresh.load_state_dict(ref_sd)
    np.testing.assert_array_equal(fresh.net.flat.data.cpu().numpy(), _flat(g, "mb64_", "after_"))
    assert fresh.opt_step == 6 and fresh.lr == pytest.approx(3e-4)
    np.testing.assert_allclose(fresh.exp_avg.cpu().numpy(), algo.exp_avg.cpu().numpy(), rtol=2e-4, atol=1e-7)
    # own round trip restores everything needed to continue bit-identically (counters, statistics, moments) -- in either
    # launch mode: eager and captured updates number their device-side permutations alike (from the optimizer step count
    # at the start of the update, PPO._device_perm), so a checkpoint of an eager run continues in a captured one and back
    for mode_src, mode_dst in ((False, False), (False, True), (True, True), (True, False)):
        src, _, buf2 = _job(g, "mb64", use_graph=mode_src)
        src.shuffle = "device"
        for _ in range(3 if mode_src else 1):  # (graph mode: eager warm-up, capture, replay)
            with policy_within_training_step(src):
                src.update(buf2, 64, 1)
        clone = PPO(net=DiscreteActorCritic(18, 5, 64, device=DEV, seed=5), shuffle="device", dispatch="pooled",
                    use_graph=mode_dst)
        clone.load_state_dict(src.state_dict())
        assert torch.equal(clone.net.flat.data, src.net.flat.data) and torch.equal(clone.exp_avg_sq, src.exp_avg_sq)
        assert clone.opt_step == src.opt_step > 0
        for _ in range(2):
            for a in (src, clone):
                with policy_within_training_step(a):
                    a.update(buf2, 64, 1)
            assert torch.equal(clone.net.flat.data, src.net.flat.data), (mode_src, mode_dst)


def test_ctde_policy_state_dict_has_the_reference_keys(golden_dir):
    ck = np.load(os.path.join(golden_dir, "checkpoint.npz"), allow_pickle=True)
    pol = CTDEPolicy(actor=DecentralizedActor(6, 3, hidden_dim=16, device=DEV, seed=0),
                     critic=CentralizedCritic(12, 2, hidden_dim=16, device=DEV, seed=1))
    sd = pol.state_dict()
    assert list(sd) == list(ck["ctde_keys"])
    assert [str(list(v.shape)) for v in sd.values()] == [s.replace(" ", "").replace(",", ", ") for s in ck["ctde_shapes"]]
    before = pol.critic.flat.data.clone()
    pol.critic.flat.data.zero_()
    pol.load_state_dict(sd)
    assert torch.equal(pol.critic.flat.data, before)
    o = pol.optim_actor.state_dict()
    assert set(o) == {"state", "param_groups"} and o["param_groups"][0]["params"] == list(range(6))
