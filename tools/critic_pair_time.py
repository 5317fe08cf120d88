"""Back-to-back time of the C3 critic launch pair (critic_rows_train + critic_dw1) and of the segmented Adam launch over the
step's slab sets (HIP events around graphs of 10 launches).  Round 4 used experiment builds of it to compare dW1 splits and
slab-load widths (profiles/r04_critic_prologue_experiment.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import ops
from tianshou_marl_amd.utils.net import FlatMLP
dev = "cuda"; D, N, H = 48, 8, 128
torch.manual_seed(0)
rows, Mr = 102400, 8192
f = FlatMLP([N * D, H, H, 1], device=dev, seed=1)
joint, ret = torch.randn(rows, N * D, device=dev), torch.randn(rows * N, device=dev)
rid = torch.randperm(rows, device=dev)[:Mr].contiguous()
cfg = ops.make_ppo_cfg(value_group=N)
ev = lambda: torch.cuda.Event(enable_timing=True)
def gtime(fn, n=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = ev(), ev(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best
img = ops.critic_w1_image(f.flat.data, N * D)
ws = {}
t = gtime(lambda: ops.critic_rows_grad_ppo(f.flat.data, joint, ret, cfg, N, H, rows=rid, Mr=Mr, ws=ws, w1_image=img))
w = next(iter(ws.values()))
t2 = gtime(lambda: ops.call("tsm_critic_rows_dw1", ops.ptr(w["dh1"]), ops.ptr(joint), N * D, ops.ptr(rid), 0, 0, 0, Mr, w["nc"], ops.ptr(w["w1"]), None, None, None, None, 0, ops.stream_ptr()))
t3 = gtime(lambda: ops.critic_rows_grad_ppo(f.flat.data, joint, ret, cfg, N, H, rows=rid, Mr=Mr, ws=ws))
ws2 = {}
t4 = gtime(lambda: ops.critic_rows_grad_ppo(f.flat.data, joint, ret, cfg, N, H, rows=rid, Mr=Mr, ws=ws2, w1_image=img, split_dw2=True))
print(f"critic pair, W1 from the fragment image: {t:.2f} us; W1 gathered: {t3:.2f} us; dW1 launch alone: {t2:.2f} us; dW1 chunk slabs {w['nc']}; "
      f"pair with dW2 in the split-K pass: {t4:.2f} us")
# segmented Adam at the C3 step's slab sets: actor 256 x 23429, dW1 64 x 49152, rest 256 x 17025
P_a, nW1, nr = 23429, 49152, 17025
n = P_a + nW1 + nr
sa, sw, sr = torch.randn(256, P_a, device=dev), torch.randn(64, nW1, device=dev), torch.randn(256, nr, device=dev)
p, m, v = torch.randn(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
sd = torch.ones(1, dtype=torch.int64, device=dev)
img = torch.empty(128 * 384, device=dev)
t = gtime(lambda: ops.adam_step_segs(p, [(sa, 0, P_a), (sw, P_a, nW1), (sr, P_a + nW1, nr)], m, v, 1, step_dev=sd))
print(f"adam_step_segs over {(sa.numel() + sw.numel() + sr.numel()) * 4 / 1e6:.1f} MB of slabs: {t:.2f} us")
ra, rc = torch.empty(P_a, device=dev), torch.empty(nr, device=dev)
t = gtime(lambda: ops.adam_step_segs(p, [(ra.view(1, -1), 0, P_a), (sw[:32], P_a, nW1), (rc.view(1, -1), P_a + nW1, nr)], m, v, 1, step_dev=sd))
print(f"adam_step_segs over one-row actor / rest segments + 32 dW1 chunk slabs (what the side reductions leave): {t:.2f} us")
