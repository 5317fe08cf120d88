#!/usr/bin/env python3
"""A/B of the two fused-update kernels: one net per workgroup (default) vs both nets in one workgroup.
Checks that gradient slabs and loss statistics are BIT-IDENTICAL and times one gradient step (update + Adam) of each.

    python tools/ab_update_variant.py
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
lib.tsm_debug_set_update_variant.argtypes = [ctypes.c_int]


def run(variant, D, A, n, M, cfg, use_image, value_clip):
    lib.tsm_debug_set_update_variant(variant)
    torch.manual_seed(0)
    net = DiscreteActorCritic(D, A, 64, device=dev, seed=0)
    P = net.flat.data
    obs = torch.randn(n, D, device=dev)
    act = torch.randint(0, A, (n,), dtype=torch.int32, device=dev)
    logp, adv, ret, v_old = (torch.randn(n, device=dev) for _ in range(4))
    logp = logp * 0.3 - 1.5
    perm = torch.randperm(n, device=dev)[:M].contiguous()
    stats = ops.ppo_adv_stats(adv, torch.tensor([0, M], device=dev), perm=perm)
    slabs, sc = ops.ppo_update_fused(P, obs, act, logp, adv, ret, cfg, A, 64, adv_stats=stats[0], perm=perm, M=M,
                                     v_s_old=v_old if value_clip else None, image=net.image if use_image else None)
    torch.cuda.synchronize()
    return slabs.clone(), sc.clone(), (P, obs, act, logp, adv, ret, stats, perm, net)


def timeit(variant, pack, cfg, M):
    lib.tsm_debug_set_update_variant(variant)
    P, obs, act, logp, adv, ret, stats, perm, net = pack
    nb = ops.ppo_update_grid(M)
    slabs = torch.empty(nb, P.numel(), device=dev)
    partial = torch.empty(nb * 4, dtype=torch.float64, device=dev)
    p_, m_, v_ = P.clone(), torch.zeros_like(P), torch.zeros_like(P)
    img = net.image.clone()

    def step():
        ops.ppo_update_fused(p_, obs, act, logp, adv, ret, cfg, 5, 64, adv_stats=stats[0], perm=perm, M=M, n_blocks=nb,
                             slabs=slabs, partial=partial, want_scalars=False, image=img)
        ops.adam_step(p_, slabs, m_, v_, 1, lr=0.0, image=img, image_map=net.image_map)

    step()
    torch.cuda.synchronize()
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
    return e0.elapsed_time(e1) * 1e3 / 200


ok = True
for (D, A, n, M, kw, use_image, vclip) in [
        (18, 5, 76800, 4096, {}, True, False), (18, 5, 76800, 5120, {}, True, False), (18, 5, 5000, 1000, {}, False, False),
        (48, 5, 9000, 4099, dict(dual_clip=2.0, value_clip=True), True, True), (33, 9, 3000, 257, dict(adv_norm=False), True, False),
        (18, 5, 76800, 4096, dict(adv_norm=False, loss_kind=1), True, False), (18, 5, 819200, 65536, {}, True, False)]:
    cfg = ops.make_ppo_cfg(**kw)
    s0, c0, pack = run(0, D, A, n, M, cfg, use_image, vclip)
    s1, c1, _ = run(1, D, A, n, M, cfg, use_image, vclip)
    same = torch.equal(s0, s1) and torch.equal(c0, c1)
    ok &= same
    line = f"D={D} A={A} M={M} {kw} image={use_image}: bit-identical={same}"
    if D == 18 and A == 5 and use_image:
        line += f"  grad step us: split {timeit(0, pack, cfg, M):.2f}  joint {timeit(1, pack, cfg, M):.2f}"
    print(line)
lib.tsm_debug_set_update_variant(0)
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
