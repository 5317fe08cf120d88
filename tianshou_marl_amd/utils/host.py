"""Host-side CPU budget of a training process.

Not part of the reference (it has no notion of a CPU quota); it exists because of a measured failure mode of the hot
loop.  PyTorch sizes its intra-op thread pool from the number of CPUs the MACHINE has (128 threads on a 256-CPU host),
not from what the process may use.  In a container with a CFS quota (cgroup `cpu.max`, 16 CPUs on the MI355X boxes) the
pool's spinning workers burn the whole quota of a 100 ms period within a few ms, the kernel then freezes EVERY thread of
the cgroup until the period ends -- including the one thread that is feeding the GPU: a 30-60 ms hole in the middle of a
hipGraph launch, once or twice per process, at a random early step (`profiles/r03_host_stall.txt`: gap between two
consecutive kernel nodes of one graph replay, `nr_throttled` of the cgroup going up; with the pool capped: no throttling,
no hole, 8 processes out of 8).

`limit_host_threads()` caps the pools to the quota; `bench.py`, the tools and the tests call it, and a training script on
a quota-limited host should too (or export OMP_NUM_THREADS)."""
from __future__ import annotations

import os
import warnings


def cpu_budget() -> int:
    """CPUs this process may actually use: min(affinity mask, cgroup CFS quota); at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def limit_host_threads(n: int | None = None) -> int:
    """Cap torch's intra-op pool (and with it OpenMP's) to `n` threads, default half the CPU budget (the other half is
    left to the HIP runtime's own threads and the interpreter), at most 16.  Returns the number set."""
    import torch

    if n is None:
        n = max(1, min(16, cpu_budget() // 2))
    if torch.get_num_threads() != n:
        torch.set_num_threads(n)
    return n


_warned = False


def warn_if_oversubscribed() -> None:
    """One warning per process when torch's pool is larger than the CPU budget (see the module text for what that costs)."""
    global _warned
    if _warned:
        return
    import torch

    budget = cpu_budget()
    if torch.get_num_threads() > budget:
        _warned = True
        warnings.warn(
            "torch uses %d intra-op threads but this process may use %d CPUs: the CFS quota can freeze the thread that feeds "
            "the GPU for tens of ms; call tianshou_marl_amd.utils.host.limit_host_threads() or set OMP_NUM_THREADS"
            % (torch.get_num_threads(), budget), RuntimeWarning, stacklevel=3)
