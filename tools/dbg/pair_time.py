"""back-to-back time of the C3 critic launch pair and of the segmented Adam launch (HIP events around graphs of 10)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = ev(), ev(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best
img = ops.critic_w1_image(f.flat.data, N * D)
for dbg, name in ((0, "4 x 96 columns, 64 chunks"), (4, "8 x 48 columns, 32 chunks"), (12, "8 x 48 columns, 64 chunks (2 per CU)"), (8, "4 x 96 columns, 128 chunks (2 per CU)")):
    ops.set_kernel_option("dbg", dbg)
    ws = {}
    t = gtime(lambda: ops.critic_rows_grad_ppo(f.flat.data, joint, ret, cfg, N, H, rows=rid, Mr=Mr, ws=ws, w1_image=img))
    w = next(iter(ws.values()))
    t2 = gtime(lambda: ops.call("tsm_critic_rows_dw1", ops.ptr(w["dh1"]), ops.ptr(joint), N * D, ops.ptr(rid), 0, 0, 0, Mr, w["nc"], ops.ptr(w["w1"]), None, 0, ops.stream_ptr()))
    print(f"critic pair with dW1 as {name}: {t:.2f} us; dw1 alone back to back {t2:.2f} us; slabs {w['nc']}")
ops.set_kernel_option("dbg", 0)
# segmented Adam at the C3 step's slab sets: actor 256 x 23429, dW1 64 x 49152, rest 256 x 17025
P_a, nW1, nr = 23429, 49152, 17025
n = P_a + nW1 + nr
sa, sw, sr = torch.randn(256, P_a, device=dev), torch.randn(64, nW1, device=dev), torch.randn(256, nr, device=dev)
p, m, v = torch.randn(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
sd = torch.ones(1, dtype=torch.int64, device=dev)
img = torch.empty(128 * 384, device=dev)
for P_a, nr in ((23429, 17025),):
  n = P_a + nW1 + nr
  sa, sw, sr = torch.randn(256, P_a, device=dev), torch.randn(64, nW1, device=dev), torch.randn(256, nr, device=dev)
  p, m, v = torch.randn(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
  for dbg in (0,):
    ops.set_kernel_option("dbg", dbg)
    t = gtime(lambda: ops.adam_step_segs(p, [(sa, 0, P_a), (sw, P_a, nW1), (sr, P_a + nW1, nr)], m, v, 1, step_dev=sd))
    print(f"adam_step_segs ({'4-B' if dbg else '16-B'} slab loads, strides {P_a} / {nr}) over {(sa.numel() + sw.numel() + sr.numel()) * 4 / 1e6:.1f} MB of slabs: {t:.2f} us")
ops.set_kernel_option("dbg", 0)
