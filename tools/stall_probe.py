#!/usr/bin/env python3
"""One bounded look at the sporadic multi-ms stall of back-to-back 4096-env collects (DESIGN.md "open observation").

    python tools/stall_probe.py [n_collects] [n_env] [host|device]

`device`: the episode record is written to HBM and copied to the pinned slot afterwards (Collector.stage_episode_record)
instead of being written to mapped pinned host memory from inside the kernel.

Every collect is bracketed by HIP events; the rollout kernel's per-workgroup start / end stamps (tsm_debug_set_stamps:
slots 64 + 2b, 65 + 2b for workgroups b < 256, 100 MHz wall clock) are read after each one.  For the slowest launches the
script prints where the time went: late workgroup STARTS (dispatch / occupancy), long workgroup DURATIONS (something inside
the kernel, e.g. the episode-record writes to mapped pinned host memory) or neither (the gap is outside the kernel:
launch path / end-of-kernel drain).  Runs once; it does not loop until a stall shows."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi  # noqa: E402
from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step  # noqa: E402
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer  # noqa: E402
from tianshou_marl_amd.data.collector import Collector  # noqa: E402
from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv  # noqa: E402
from tianshou_marl_amd.utils.net import DiscreteActorCritic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = "cuda"
    env = DeviceSimpleSpreadVectorEnv(E, 3, device=dev, seed=1)
    net = DiscreteActorCritic(18, 5, 64, device=dev, seed=0)
    algo = PPO(net=net)
    buf = DeviceVectorReplayBuffer(E * 25, E, 3, 18, device=dev)
    col = Collector(algo, env, buf)
    record = sys.argv[3] if len(sys.argv) > 3 else "host"
    col.stage_episode_record = record == "device"
    col.reset()
    st = torch.zeros(1024, dtype=torch.int64, device=dev)
    lib = _abi.load()
    lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]
    lib.tsm_debug_set_stamps(st.data_ptr())
    n_wg = min(256, -(-E // 5))
    times, rows = [], []
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    with policy_within_training_step(algo):
        for i in range(n):
            e0, e1 = ev(), ev()
            e0.record()
            col.collect(n_step=E * 25)
            e1.record()
            torch.cuda.synchronize()
            col.reset_buffer(keep_statistics=True)
            dt = e0.elapsed_time(e1) * 1e3
            blk = st.cpu().numpy()[64:64 + 2 * n_wg].reshape(-1, 2).astype(np.float64) / 100.0  # us
            t0 = blk[:, 0].min()
            rows.append(dict(i=i, event_us=round(dt, 1), start_spread_us=round(blk[:, 0].max() - t0, 1),
                             wg_dur_med_us=round(float(np.median(blk[:, 1] - blk[:, 0])), 1),
                             wg_dur_max_us=round(float((blk[:, 1] - blk[:, 0]).max()), 1),
                             span_first256_us=round(blk[:, 1].max() - t0, 1)))
            times.append(dt)
    lib.tsm_debug_set_stamps(None)
    t = np.array(times[5:])
    print(json.dumps(dict(n=n, n_env=E, record=record, p50_us=round(float(np.median(t)), 1), p99_us=round(float(np.percentile(t, 99)), 1),
                          max_us=round(float(t.max()), 1), n_over_2x_median=int((t > 2 * np.median(t)).sum()))))
    for r in sorted(rows[5:], key=lambda r: -r["event_us"])[:6]:
        print(json.dumps(r))
    print("typical:", json.dumps(rows[len(rows) // 2]))


if __name__ == "__main__":
    main()
