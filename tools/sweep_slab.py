#!/usr/bin/env python3
"""Sweep of the headline gradient step (fused update + Adam) over the slab count and the slab store flavour.

    python tools/sweep_slab.py

For M = 4096 and 5120 rows (the two minibatch sizes of the headline update): grid of n_blocks workgroup pairs, each
accumulating ceil(tiles / n_blocks) 16-row tiles in registers before it writes ONE 44.6 KB gradient slab, times the
store flavour (non-temporal | plain write-back | agent-scope write-through).  Every cell: microseconds per gradient step
(update + Adam, 20 steps per graph replay, 10 replays), slab bytes written + read back per step, and whether two runs of
the same cell give bit-identical slabs.
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi, ops  # noqa: E402
from tianshou_marl_amd.utils.net import DiscreteActorCritic  # noqa: E402

dev = "cuda"
lib = _abi.load()
lib.tsm_debug_set_slab_store.argtypes = [ctypes.c_int]
STORES = {0: "non-temporal", 1: "write-back", 2: "write-through"}


def setup(M, n=76800, D=18, A=5):
    torch.manual_seed(0)
    net = DiscreteActorCritic(D, A, 64, device=dev, seed=0)
    obs = torch.randn(n, D, device=dev)
    act = torch.randint(0, A, (n,), dtype=torch.int32, device=dev)
    logp, adv, ret = (torch.randn(n, device=dev) for _ in range(3))
    logp = logp * 0.3 - 1.5
    perm = torch.randperm(n, device=dev)[:M].contiguous()
    stats = ops.ppo_adv_stats(adv, torch.tensor([0, M], device=dev), perm=perm)
    return net, obs, act, logp, adv, ret, stats, perm


def cell(pack, M, nb, cfg):
    net, obs, act, logp, adv, ret, stats, perm = pack
    P = net.flat.data
    slabs = torch.zeros(nb, P.numel(), device=dev)
    partial = torch.empty(nb * 4, dtype=torch.float64, device=dev)
    p_, m_, v_ = P.clone(), torch.zeros_like(P), torch.zeros_like(P)
    img = net.image.clone()

    def step():
        ops.ppo_update_fused(p_, obs, act, logp, adv, ret, cfg, 5, 64, adv_stats=stats[0], perm=perm, M=M, n_blocks=nb,
                             slabs=slabs, partial=partial, want_scalars=False, image=img)
        ops.adam_step(p_, slabs, m_, v_, 1, lr=0.0, image=img, image_map=net.image_map)

    step()
    torch.cuda.synchronize()
    first = slabs.clone()
    step()
    torch.cuda.synchronize()
    same = torch.equal(first, slabs)
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(20):
            step()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 200, same, first.sum(0)


def main():
    cfg = ops.make_ppo_cfg()
    for M in (4096, 5120):
        pack = setup(M)
        tiles = -(-M // 16)
        ref = None
        print(f"M={M} ({tiles} tiles), default grid {ops.ppo_update_grid(M)}")
        for nb in sorted({tiles, (tiles + 1) // 2, (tiles + 2) // 3, (tiles + 3) // 4}, reverse=True):
            row = []
            for st in STORES:
                lib.tsm_debug_set_slab_store(st)
                us, same, total = cell(pack, M, nb, cfg)
                ref = total if ref is None else ref
                err = float((total - ref).abs().max() / ref.abs().max())
                row.append(f"{STORES[st]} {us:6.2f} us{'' if same else ' NOT-REPRODUCIBLE'} (rel {err:.1e})")
            mb = 2 * nb * pack[0].flat.numel() * 4 / 1e6
            print(f"  n_blocks {nb:4d}  slab write+read {mb:5.1f} MB | " + " | ".join(row), flush=True)
    lib.tsm_debug_set_slab_store(0)


if __name__ == "__main__":
    main()
