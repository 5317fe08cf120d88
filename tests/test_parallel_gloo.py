"""World-size-2 and world-size-8 `gloo` tests (CPU) of the env-sharded data-parallel host logic (tianshou_marl_amd/parallel.py).

Covers what the N>1 path adds over N=1: the env shard arithmetic, the replica broadcast at attach time, and the
flat-gradient mean all-reduce that keeps replicas identical.  The gradient itself comes from the HIP kernels
(GPU tests); here each rank supplies a known vector and applies a plain Adam step in numpy so that the
"replicas stay bit-identical" property is checked end to end across two real processes.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tianshou_marl_amd.parallel import GradSync, attach_data_parallel, learn_lockstep, shard_range


def test_shard_range_partitions_every_env_once():
    for n, w in [(1024, 8), (10, 3), (7, 8), (4096, 5)]:
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n


class _Net:
    def __init__(self, flat):
        self.flat = torch.nn.Parameter(flat, requires_grad=False)


class _Algo:
    """The attributes attach_data_parallel touches on a PPO object."""

    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        self.net = _Net(torch.randn(257, generator=g))
        self.exp_avg = torch.randn(257, generator=g)
        self.exp_avg_sq = torch.rand(257, generator=g)
        self._grad_sync = None


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        algo = _Algo(seed=100 + rank)  # replicas start DIFFERENT: attach must make them equal to rank 0
        sync = attach_data_parallel(algo, dist)
        assert isinstance(sync, GradSync) and sync.world == world and sync.rank == rank and algo._grad_sync is sync
        ref = _Algo(seed=100)
        assert torch.equal(algo.net.flat.data, ref.net.flat.data)
        assert torch.equal(algo.exp_avg, ref.exp_avg) and torch.equal(algo.exp_avg_sq, ref.exp_avg_sq)
        # env shards: every rank owns a different slice, the union is the whole job
        lo, hi = shard_range(10, world, rank)
        owned = torch.zeros(10)
        owned[lo:hi] = 1
        dist.all_reduce(owned)
        assert torch.equal(owned, torch.ones(10))
        # three "gradient steps": rank-local gradients differ, the synced mean and the update do not
        p = algo.net.flat.data
        m, v = algo.exp_avg, algo.exp_avg_sq
        for step in range(1, 4):
            g = torch.full((257,), float(rank + 1)) * step + torch.arange(257) * 1e-3
            sync.all_reduce_mean_(g)
            expect = torch.full((257,), (1 + world) / 2.0) * step + torch.arange(257) * 1e-3
            assert torch.allclose(g, expect, rtol=0, atol=2e-5)
            m.mul_(0.9).add_(g, alpha=0.1)
            v.mul_(0.999).addcmul_(g, g, value=0.001)
            p.sub_(1e-3 * (m / (1 - 0.9 ** step)) / ((v / (1 - 0.999 ** step)).sqrt() + 1e-8))
        # pre-scaled contributions + plain sum == mean (the form the update path uses: the 1/world factor is applied
        # inside the slab-reduction kernel)
        g = (torch.full((9,), float(rank + 1)) + torch.arange(9)) / world
        sync.all_reduce_sum_(g)
        assert torch.allclose(g, torch.full((9,), (1 + world) / 2.0) + torch.arange(9), rtol=0, atol=2e-5)
        # lock-step training of two policy groups (parallel.learn_lockstep): the flat gradients that fall due at the same
        # gradient step are packed into ONE all-reduce; a group with more steps goes on alone.  Generators stand in for
        # PPO.learn_steps: they yield rank-local gradients (pre-scaled by 1 / world) and read them back reduced.
        reduces = []
        orig = sync.all_reduce_sum_
        sync.all_reduce_sum_ = lambda t: (reduces.append(t.numel()), orig(t))[1]

        def group(n_param, n_steps, base):
            seen = []
            for k in range(n_steps):
                g = torch.full((n_param,), (base + k) * (rank + 1) / world)
                yield g
                seen.append(g.clone())
            return seen

        got_a, got_b = learn_lockstep([group(5, 2, 1.0), group(7, 3, 10.0)], sync)
        sync.all_reduce_sum_ = orig
        assert reduces == [12, 12, 7]  # steps 1-2: both groups in one packed buffer; step 3: group b alone
        mean_rank = (1 + world) / 2.0  # sum over ranks of (rank + 1) / world
        for k, g in enumerate(got_a):
            assert torch.allclose(g, torch.full((5,), (1.0 + k) * mean_rank))
        for k, g in enumerate(got_b):
            assert torch.allclose(g, torch.full((7,), (10.0 + k) * mean_rank))
        # the same lock-step as a generator of its collectives with a PREALLOCATED packing buffer (what the captured form,
        # parallel.learn_lockstep_graph, replays: fixed addresses) -- f32 gradients go through the buffer, a step whose
        # tensors are f64 statistics packs (or mixed) is packed in the widest type; results as above
        from tianshou_marl_amd.parallel import lockstep_steps

        def group2(n_param, base):
            pack = torch.full((3,), base * (rank + 1) / world, dtype=torch.float64)  # the advantage-statistics pack comes first
            yield pack
            g = torch.full((n_param,), (base + 1) * (rank + 1) / world)
            yield g
            return pack.clone(), g.clone()

        buf = torch.empty(12)
        res = [None, None]
        seen_t = []
        for t in lockstep_steps([group2(5, 1.0), group2(7, 10.0)], res, buf):
            seen_t.append((t.dtype, t.numel(), t.data_ptr() == buf.data_ptr()))
            sync.all_reduce_sum_(t)
        assert seen_t == [(torch.float64, 6, False), (torch.float32, 12, True)]
        for (pack, g), base in zip(res, (1.0, 10.0)):
            assert torch.allclose(pack, torch.full((3,), base * mean_rank, dtype=torch.float64))
            assert torch.allclose(g, torch.full((g.numel(),), (base + 1) * mean_rank))
        # the captured lock-step selects its graph by (groups, row counts): ranks that disagree on them are stopped by the same
        # every-call agreement check before any graph is replayed
        sync.check_same(["adversaries:300:None:1:0", "good:300:None:1:0"], "the policy groups that train in this step (and their row counts)")
        with pytest.raises(ValueError, match="disagree"):
            sync.check_same([f"adversaries:{300 + rank}:None:1:0"], "the policy groups that train in this step (and their row counts)")
        # every rank must take the same number of gradient steps per update (unequal env shards can split into a
        # different number of minibatches -> a different number of all-reduces -> deadlock): agreed values pass, a
        # disagreement raises on EVERY rank instead of hanging
        sync.require_equal(18, "the number of gradient steps per update")
        sync.require_equal(18, "the number of gradient steps per update")
        with pytest.raises(ValueError, match="disagree"):
            sync.require_equal(30 + rank, "the number of gradient steps per update")
        # ... also when ONE rank comes back with a count all ranks have agreed on before (n_episode collection: the
        # number of valid rows varies per rank and per update): the check is a collective on every call, never skipped
        # because of what this rank has seen
        with pytest.raises(ValueError, match="disagree"):
            sync.require_equal(18 if rank == 0 else 12, "the number of gradient steps per update")
        # advantage statistics of the GLOBAL minibatch: every rank holds (mean, unbiased std) of its own
        # part of each minibatch; pack -> ONE all-reduce -> unpack turns them into the statistics of the union, identical
        # on every rank.  The pack / unpack arithmetic is a pair of HIP kernels in the product (checked bit for bit in
        # tests/test_gpu_parallel.py); here the same formulas in torch stand in for them and the PROTOCOL is under test.
        def pack_fn(stats, mb_start):
            n = (mb_start[1:] - mb_start[:-1]).double()
            m, sd = stats[:, 0].double(), stats[:, 1].double()
            return torch.stack([n, n * m, (n - 1.0) * (sd * sd) + n * m * m], dim=1).contiguous()

        def unpack_fn(pack, stats):
            m = pack[:, 1] / pack[:, 0]
            var = (pack[:, 2] - pack[:, 0] * m * m) / (pack[:, 0] - 1.0)
            stats.copy_(torch.stack([m, var.clamp_min(0.0).sqrt()], dim=1).float())

        sync._stat_codec = (pack_fn, unpack_fn)
        gen = torch.Generator().manual_seed(7)
        parts = [[torch.randn(n, generator=gen) * (1 + k) + k for n in (40 + 8 * k, 64)] for k in range(world)]  # [rank][minibatch]
        mine = torch.tensor([[float(x.mean()), float(x.std())] for x in parts[rank]], dtype=torch.float32)
        mb_start = torch.tensor([0, parts[rank][0].numel(), parts[rank][0].numel() + parts[rank][1].numel()])
        sync.merge_adv_stats_(mine, mb_start)
        for j in range(2):
            union = torch.cat([parts[k][j] for k in range(world)]).double()
            assert mine[j, 0].item() == pytest.approx(float(union.mean()), rel=1e-6, abs=1e-6)
            assert mine[j, 1].item() == pytest.approx(float(union.std()), rel=1e-6)
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert all(torch.equal(both[0], b) for b in both)  # bit-identical on every rank
        sync.global_adv_stats = False                       # rank-local statistics: no collective, untouched
        local = torch.tensor([[1.0, 2.0]])
        assert torch.equal(sync.merge_adv_stats_(local, torch.tensor([0, 5])), torch.tensor([[1.0, 2.0]]))
        sync.global_adv_stats = True
        np.save(os.path.join(out_dir, f"p{rank}.npy"), p.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8])
def test_replicas_stay_identical(tmp_path, world):
    """world 8 = the node the scaling run uses (BASELINE configs[3] / [4]): the lock-step generator and its packed-reduce layout,
    the every-call agreement checks (a disagreement on the step count or on the groups raises on EVERY rank, no hang), the
    merged advantage statistics and the replicas' bit-identity, with eight real processes."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ps = [np.load(tmp_path / f"p{r}.npy") for r in range(world)]
    assert all(np.array_equal(ps[0], p) for p in ps[1:])  # bit-identical replicas after synced steps
