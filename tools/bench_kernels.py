#!/usr/bin/env python3
"""Kernel micro-benchmarks (roofline grid of SURVEY.md 8d): GAE and PPO-loss kernels.

    python tools/bench_kernels.py [--iters 50]
Prints one JSON line per case: algorithmic bytes / measured kernel time vs the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import ops  # noqa: E402

PEAK = 8.0e12


def timeit(fn, iters, warmup=5):
    # the launches are captured into one hipGraph so that host launch overhead (python + ctypes, ~10 us) cannot
    # hide a 5 us kernel; best of 3 replays
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e-3 / iters)
    return best


def gae_sweep(iters):
    """Tuning grid of the GAE kernel: lanes per thread x steps per thread x waves per workgroup."""
    from tianshou_marl_amd import _abi

    lib = _abi.load()
    dev = "cuda"
    for (n_env, N, T) in [(4096, 8, 25), (1024, 3, 25), (4096, 8, 2048)]:
        L = n_env * N
        v_s, v_n, rew = (torch.randn(T, L, device=dev) for _ in range(3))
        te = (torch.rand(T, L, device=dev) < 0.01).to(torch.uint8)
        tr = torch.zeros(T, L, dtype=torch.uint8, device=dev)
        out = (torch.empty(T, L, device=dev), torch.empty(T, L, device=dev))
        ref = None
        for vec in (1,):  # VEC=2/4 (8-/16-B loads per lane) measured 3-10x slower: see profiles/r01_gae_sweep.txt
            for ch in (2, 4, 8):
                for w in (0, 2, 4, 8, 16):
                    if w and w * ch < 8 and T > 64:
                        continue
                    lib.tsm_debug_gae_config(vec, ch, w)
                    sec = timeit(lambda: ops.gae_lanes(v_s, v_n, rew, te, tr, out=out), iters)
                    if ref is None:
                        ref = (out[0].clone(), out[1].clone())
                    same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
                    byts = 22 * T * L
                    print(json.dumps(dict(kernel="gae_sweep", n_env=n_env, n_agent=N, T=T, vec=vec, ch=ch, w=w,
                                          us=round(sec * 1e6, 2), frac=round(byts / sec / PEAK, 4), same=same)), flush=True)
        lib.tsm_debug_gae_config(0, 0, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--gae-sweep", action="store_true")
    a = ap.parse_args()
    if a.gae_sweep:
        gae_sweep(a.iters)
        return
    dev = "cuda"
    for (n_env, N, T) in [(1024, 3, 25), (4096, 8, 25), (4096, 8, 256), (4096, 8, 2048)]:
        L = n_env * N
        v_s, v_n, rew = (torch.randn(T, L, device=dev) for _ in range(3))
        te = (torch.rand(T, L, device=dev) < 0.01).to(torch.uint8)
        tr = torch.zeros(T, L, dtype=torch.uint8, device=dev)
        out = (torch.empty(T, L, device=dev), torch.empty(T, L, device=dev))
        sec = timeit(lambda: ops.gae_lanes(v_s, v_n, rew, te, tr, out=out), a.iters)
        byts = 22 * T * L
        print(json.dumps(dict(kernel="gae_lanes", n_env=n_env, n_agent=N, T=T, us=sec * 1e6,
                              GBps=byts / sec / 1e9, frac=byts / sec / PEAK, bytes=byts)))
        del v_s, v_n, rew, te, tr, out
    cfg = ops.make_ppo_cfg()
    for M in (4096, 76800, 819200, 8192000):
        A = 5
        logits = torch.randn(M, A, device=dev)
        value, logp_old, adv, ret = (torch.randn(M, device=dev) for _ in range(4))
        act = torch.randint(0, A, (M,), dtype=torch.int32, device=dev)
        stats = ops.ppo_adv_stats(adv, torch.tensor([0, M], device=dev))
        sec = timeit(lambda: ops.ppo_loss_fwd_bwd(logits, value, act, logp_old, adv, ret, cfg, adv_stats=stats[0]), a.iters)
        byts = 64 * M
        print(json.dumps(dict(kernel="ppo_loss_fwd_bwd(+finalize,+allocs)", M=M, us=sec * 1e6, GBps=byts / sec / 1e9,
                              frac=byts / sec / PEAK, bytes=byts)))


if __name__ == "__main__":
    main()
