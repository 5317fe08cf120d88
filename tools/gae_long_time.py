import sys, torch
sys.path.insert(0, '.')
from tianshou_marl_amd import ops
dev = "cuda"
for T, L in [(12800, 1), (12800, 2), (25600, 4), (6000, 1)]:
    v_s, v_n, rew = (torch.randn(T, L, device=dev) for _ in range(3))
    term = torch.rand(T, L, device=dev) < 0.02
    trunc = torch.zeros(T, L, dtype=torch.bool, device=dev)
    res = {}
    for dbg in (0, 32, 64):
        ops.set_kernel_option("dbg", dbg)
        for _ in range(5): out = ops.gae_lanes(v_s, v_n, rew, term, trunc)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): out = ops.gae_lanes(v_s, v_n, rew, term, trunc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); e0.record(); [g.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
        res[dbg] = (e0.elapsed_time(e1) * 1e3 / 100, out[1].clone())
    ops.set_kernel_option("dbg", 0)
    print(T, L, "CH16 %.2f us  CH8 %.2f us  CH4 %.2f us  max|diff| %.3g %.3g" % (res[0][0], res[32][0], res[64][0], (res[0][1] - res[32][1]).abs().max().item(), (res[0][1] - res[64][1]).abs().max().item()))
