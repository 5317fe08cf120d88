#!/usr/bin/env python3
"""VERDICT r4 item 1, measured: what a gradient step split as  kernel A (forward + loss + input-gradient backward, activations
published) + kernel B (weight gradients as a K = M product per parameter tile + Adam in place)  could cost at the headline
minibatch (4096 rows, obs 18, 64-wide nets), against the step as it is (update kernel with register-resident weight gradients +
256 slabs, then adam_kernel).

Kernel A exists as a timing variant of the update kernel (tsm_debug_set_update_variant(2): no gradient comes out of it).  Kernel B
is bounded from below by the optimizer launch on an already reduced gradient (ONE slab): whatever B does, it also loads p / m / v,
steps and stores every parameter and the weight image.  Graphs of 20 launches, HIP events, min over 5 replays.

    python tools/ab_kernel_a.py
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
D, A, n, M = 18, 5, 76800, 4096
torch.manual_seed(0)
net = DiscreteActorCritic(D, A, 64, device=dev, seed=0)
P = net.flat.data
obs = torch.randn(n, D, device=dev)
act = torch.randint(0, A, (n,), dtype=torch.int32, device=dev)
logp, adv, ret = (torch.randn(n, device=dev) for _ in range(3))
logp = logp * 0.3 - 1.5
perm = torch.randperm(n, device=dev)[:M].contiguous()
cfg = ops.make_ppo_cfg()
stats = ops.ppo_adv_stats(adv, torch.tensor([0, M], device=dev), perm=perm)
nb = ops.ppo_update_grid(M)
slabs = torch.empty(nb, P.numel(), device=dev)
one = torch.randn(1, P.numel(), device=dev) * 1e-3
partial = torch.empty(nb * 4, dtype=torch.float64, device=dev)
p_, m_, v_ = P.clone(), torch.zeros_like(P), torch.zeros_like(P)
img = net.image.clone()


def update():
    ops.ppo_update_fused(p_, obs, act, logp, adv, ret, cfg, A, 64, adv_stats=stats[0], perm=perm, M=M, n_blocks=nb, slabs=slabs,
                         partial=partial, want_scalars=False, image=img)


def adam(g):
    ops.adam_step(p_, g, m_, v_, 1, lr=0.0, image=img, image_map=net.image_map)


def timed(fn, n_launch=20, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(n_launch):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n_launch)
    return best


res = {}
lib.tsm_debug_set_update_variant(0)
res["update kernel as it is (weight gradients in registers, 256 slabs of 44.6 KB)"] = timed(update)
res["... + adam_kernel over the 256 slabs = the gradient step as it is"] = timed(lambda: (update(), adam(slabs)))
res["adam_kernel over 256 slabs, alone"] = timed(lambda: adam(slabs))
res["adam_kernel over ONE slab, alone (lower bound of kernel B)"] = timed(lambda: adam(one))
lib.tsm_debug_set_update_variant(2)
res["kernel A (no weight gradients; H1 / H2 / dH2 / dH1 / X / dOut published: 19.7 KB per tile and net)"] = timed(update)
res["kernel A + adam_kernel over ONE slab = lower bound of the A / B gradient step"] = timed(lambda: (update(), adam(one)))
lib.tsm_debug_set_update_variant(0)
print("# tools/ab_kernel_a.py (MI355X): us per launch (pair), graphs of 20, HIP events, min of 5 replays; M = 4096 rows, obs 18, 64-wide nets")
for k, v in res.items():
    print(f"{v:8.2f} us  {k}")
a, b = res["... + adam_kernel over the 256 slabs = the gradient step as it is"], res["kernel A + adam_kernel over ONE slab = lower bound of the A / B gradient step"]
print(f"# room for kernel B's own work (K = 4096 products for 11 142 parameters, LDS fold, in-launch reduction if K is split over "
      f"workgroups): {a - b:.2f} us before the A / B step is slower than the step as it is")
