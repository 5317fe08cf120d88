#!/usr/bin/env python3
"""Phase time stamps of the one-launch actor step at BASELINE configs[2]: 65 536 samples of 48 -> 128 -> 128 -> 5, workgroup 0,
its tiles 1-3 (100 MHz wall clock).  The minibatch size picks the kernel (csrc/actor_rows64.hip: 64-sample tiles, W2 in
registers; csrc/ppo_rows.hip: 32-sample tiles); TSM_ACTOR_TILE=32 | 64 forces one.

    python tools/stamp_actor_rows.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi, ops  # noqa: E402
from tianshou_marl_amd.utils.net import MLPActorCritic  # noqa: E402

dev = "cuda"
D, A, H, n, M = 48, 5, 128, 819200, 65536
torch.manual_seed(0)
net = MLPActorCritic(D, A, (H, H), critic_obs_dim=8 * D, device=dev, seed=1)
obs = torch.randn(n, D, device=dev)
act = torch.randint(0, A, (n,), dtype=torch.int32, device=dev)
lp, adv = torch.randn(n, device=dev) * 0.3 - 1.5, torch.randn(n, device=dev)
perm = torch.randperm(n, device=dev)[:M].contiguous()
stats = ops.ppo_adv_stats(adv, torch.tensor([0, M], device=dev), perm=perm, max_rows=M)
cfg = ops.make_ppo_cfg()
nb = ops.ppo_actor_rows_grid(M)
slabs = torch.empty(nb, net.n_actor, device=dev)
part = torch.empty(nb * 4, dtype=torch.float64, device=dev)
st = torch.zeros(2048, dtype=torch.int64, device=dev)
lib = _abi.load()
lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.ppo_actor_rows_update(net.actor.flat.data, obs, act, lp, adv, cfg, A, H, adv_stats=stats[0], perm=perm, M=M, n_blocks=nb,
                              slabs=slabs, partial=part)
# the launch back to back, without stamps: a graph of 10 launches, best of 5 replays
g = torch.cuda.CUDAGraph()
with ops.graph_capture(g):
    for _ in range(10):
        ops.ppo_actor_rows_update(net.actor.flat.data, obs, act, lp, adv, cfg, A, H, adv_stats=stats[0], perm=perm, M=M, n_blocks=nb,
                                  slabs=slabs, partial=part)
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 100.0)
for _ in range(1000):   # ~1 s of the same launch back to back: the stamped launch below runs at the clock the part settles to under this load
    g.replay()
lib.tsm_debug_set_stamps(st.data_ptr())
ops.ppo_actor_rows_update(net.actor.flat.data, obs, act, lp, adv, cfg, A, H, adv_stats=stats[0], perm=perm, M=M, n_blocks=nb,
                          slabs=slabs, partial=part)
torch.cuda.synchronize()
if os.environ.get("STAMP_IN_GRAPH") == "1":   # the same stamps from the LAST of 2 000 launches replayed back to back (graph of 10 x 200)
    g2 = torch.cuda.CUDAGraph()
    with ops.graph_capture(g2):
        for _ in range(10):
            ops.ppo_actor_rows_update(net.actor.flat.data, obs, act, lp, adv, cfg, A, H, adv_stats=stats[0], perm=perm, M=M, n_blocks=nb,
                                      slabs=slabs, partial=part)
    for _ in range(200):
        g2.replay()
    torch.cuda.synchronize()
    print("(stamps of the last launch of a graph replayed back to back)")
lib.tsm_debug_set_stamps(None)
sa = st.cpu().numpy()
if sa[201] > sa[200] > 0:  # (the 64-sample kernel stamps its prologue and epilogue too)
    print(f"prologue (weights -> LDS / registers, first ids and rows) {(sa[201] - sa[200]) / 100.0:.2f} us; slab + statistics after the last tile "
          f"{(sa[203] - sa[202]) / 100.0:.2f} us; kernel entry -> exit of workgroup 0 {(sa[203] - sa[200]) / 100.0:.2f} us")
if sa[205] > sa[204] > 0:
    print(f"shader clock over the tile loop of workgroup 0: {(sa[205] - sa[204]) / (sa[202] - sa[201]) * 100.0:.0f} MHz (s_memtime ticks per 100 MHz s_memrealtime tick)")
wg = sa[1024:1024 + 2 * nb].reshape(nb, 2).astype("float64") / 100.0
if nb <= 512 and wg[:, 0].min() > 0:   # start / end of every workgroup (64-sample kernel)
    t0 = wg[:, 0].min()
    life = wg[:, 1] - wg[:, 0]
    print(f"workgroups: first start -> last end {wg[:, 1].max() - t0:.2f} us; starts spread over {wg[:, 0].max() - t0:.2f} us (median {float(sorted(wg[:, 0] - t0)[nb // 2]):.2f}); "
          f"lifetime min / median / max {life.min():.2f} / {float(sorted(life)[nb // 2]):.2f} / {life.max():.2f} us")
    by8 = [f"{(wg[x::8, 0] - t0).mean():.2f}/{life[x::8].mean():.2f}" for x in range(8)]
    print("  mean start / lifetime by blockIdx % 8 (= XCD): " + ", ".join(by8))
s = sa[:64].reshape(4, 16)
names = ["P0 commit X", "P1 layer 1", "P2 layer 2", "P3 logits", "P4 loss head (16 lanes per sample)", "P5 dW3 + dH2 mfma", "dH2 write",
         "P6 dW2 + dH1 mfma", "dH1 write", "P7 dW1"]
for it in range(1, 4):
    d = [(s[it][k + 1] - s[it][k]) / 100.0 for k in range(10)]
    print(f"tile {it}: total {(s[it][10] - s[it][0]) / 100.0:.2f} us   " + ", ".join(f"{n_} {x:.2f}" for n_, x in zip(names, d)))
for base, who in ((0, "wave 0"), (512, "wave 7")):
    f = sa[base:base + 64].reshape(4, 16)
    if f[1][15] > f[1][11] > 0:   # finer stamps (only in a diagnostic build of actor_rows64.hip)
        for it in range(1, 4):
            t = lambda k: f[it][k] / 100.0
            print(f"  {who} tile {it}: P2 stream {t(11) - t(2):.2f}, epilogue {t(12) - t(11):.2f}, barrier {t(3) - t(12) if base == 0 else float('nan'):.2f} | "
                  f"P6 dW2 stream {t(13) - t(7):.2f}, bias sums {t(14) - t(13):.2f}, dH1 stream {t(15) - t(14):.2f}, barrier "
                  f"{t(8) - t(15) if base == 0 else float('nan'):.2f}   (wave-7 phase starts relative to wave 0: P2 {f[it][2] / 100.0 - sa[it * 16 + 2] / 100.0:+.2f}, "
                  f"P6 {f[it][7] / 100.0 - sa[it * 16 + 7] / 100.0:+.2f})")
rows = 64 if -(-M // 64) >= ops.device_info()["n_cu"] else 32  # (the rule of tsm_ppo_actor_rows_grid)
if os.environ.get("TSM_ACTOR_TILE") in ("32", "64"):
    rows = int(os.environ["TSM_ACTOR_TILE"])
print(f"back to back: {best:.2f} us per launch (graph of 10, best of 5)")
print(f"n_blocks {nb}, {rows}-sample tiles, {M // rows // nb} tiles per workgroup (P3: logits on {'all eight waves, two k halves' if rows == 64 else 'two waves'})")
