#!/usr/bin/env python3
"""Duration of tsm_gae_lanes on few long lanes (csrc/gae.hip: gae_long_kernel / gae_long_par_kernel): super-chunks of 4096 steps in
turn on one workgroup per lane vs side by side on different workgroups (scan workspace registered).  Graph of 20 launches, HIP events.
(Round 4 history: with 16 steps per thread instead of 4 the one-workgroup form spilled 404 registers -- 122.6 us for 12 800 x 1.)

    python tools/gae_long_time.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi, ops  # noqa: E402

dev = "cuda"
ops.ensure_scan_workspace(dev)
nbytes = int(_abi.call("tsm_gae_scan_workspace_bytes"))
for T, L in [(12800, 1), (12800, 2), (25600, 4), (6000, 1), (100000, 1)]:
    v_s, v_n, rew = (torch.randn(T, L, device=dev) for _ in range(3))
    term = torch.rand(T, L, device=dev) < 0.02
    trunc = torch.zeros(T, L, dtype=torch.bool, device=dev)
    res = {}
    for par in (False, True):
        _abi.call("tsm_gae_set_scan_workspace", ops._scan_ws.data_ptr() if par else None, nbytes if par else 0)
        for _ in range(5):
            out = ops.gae_lanes(v_s, v_n, rew, term, trunc)
        g = torch.cuda.CUDAGraph()
        with ops.graph_capture(g):
            for _ in range(20):
                out = ops.gae_lanes(v_s, v_n, rew, term, trunc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); e0.record(); [g.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
        res[par] = (e0.elapsed_time(e1) * 1e3 / 100, out[1].clone())
    _abi.call("tsm_gae_set_scan_workspace", ops._scan_ws.data_ptr(), nbytes)
    print(f"{T:7d} steps x {L} lanes: in turn {res[False][0]:7.2f} us, side by side {res[True][0]:7.2f} us, identical: {torch.equal(res[False][1], res[True][1])}")
