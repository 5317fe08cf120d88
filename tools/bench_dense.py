#!/usr/bin/env python3
"""Per-shape timing of the tiled f32-MFMA GEMM kernels (csrc/dense.hip): forward, dgrad+wgrad (backward)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import ops  # noqa: E402
from tianshou_marl_amd.utils.net import FlatMLP  # noqa: E402

PEAK = 157.3e12


def gtime(fn, n=10, reps=3):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e-3 / n)
    return best


def main():
    dev = "cuda"
    for B, K, O in [(102400, 384, 128), (102400, 128, 128), (102400, 128, 8), (819200, 48, 128), (32768, 48, 128),
                    (4096, 4096, 4096)]:
        net = FlatMLP([K, O], act="relu", device=dev, seed=0)
        x = torch.randn(B, K, device=dev)
        d = torch.randn(B, O, device=dev)
        t_f = gtime(lambda: net(x, save=False))
        net(x)
        t_b = gtime(lambda: net.backward(d))  # single layer: wgrad only
        flop = 2.0 * B * K * O
        print(json.dumps(dict(B=B, K=K, O=O, fwd_us=round(t_f * 1e6, 1), fwd_TF=round(flop / t_f / 1e12, 1),
                              fwd_frac=round(flop / t_f / PEAK, 3), wgrad_us=round(t_b * 1e6, 1),
                              wgrad_TF=round(flop / t_b / 1e12, 1), wgrad_frac=round(flop / t_b / PEAK, 3))), flush=True)


if __name__ == "__main__":
    main()
