"""`python bench.py --gpus N` starts N ranks itself (the reference has no launcher, only the
single-process DataParallel wrappers of utils/net/common.py:477-519).

CPU part: the launcher starts one child per rank before anything touches a GPU, propagates a failing rank as a non-zero
exit, and refuses a `--gpus` that disagrees with the launcher's WORLD_SIZE.  GPU part (`-m gpu`): the whole bench as two
real processes on one card (gloo; RCCL refuses two ranks on one device) reports the rank count the process group
counted and bit-identical replicas."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def _no_gpu() -> bool:
    import torch

    return torch.cuda.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="the failure path needs a host without a GPU")
def test_launcher_starts_every_rank_and_propagates_failure():
    r = _run(["--gpus", "3", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert r.stdout.strip() == ""  # no JSON line from a failed job
    # every started rank reports for itself that it has no GPU (ranks stopped by the launcher may not get that far)
    assert r.stderr.count("bench.py needs a GPU") >= 1
    assert "stopped the other ranks" in r.stderr


def test_gpus_must_match_the_launchers_world_size():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and "--gpus 4" in r.stderr
    r = _run([], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})  # --gpus defaults to 1
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_single_gpu_workloads_refuse_more_ranks():
    r = _run(["--gpus", "2", "--workload", "c3ppo"])
    assert r.returncode != 0 and "single-GPU job" in r.stderr


@pytest.mark.parametrize("var,val", [("TSM_DBG", "1"), ("TSM_GENERIC_KERNELS", "1"), ("TSM_ACTOR_TILE", "64"), ("TSM_ROLLOUT_FORM", "1"),
                                     ("TSM_SPLIT_BF16", "1"), ("TSM_UPDATE_MAX_BLOCKS", "128")])
def test_bench_refuses_kernel_selection_switches_before_touching_the_gpu(var, val):
    """a measurement must prove its configuration.  With any kernel-selection / diagnostics switch set in the
    environment bench.py exits non-zero BEFORE anything touches the GPU (here: a host without one -- the message is the refusal, not
    "needs a GPU") and prints no JSON line; `--allow-options` lifts the refusal (and the run then goes on to need a GPU)."""
    r = _run(["--steps", "1", "--warmup", "0"], {var: val})
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "refusing to run with " + var in r.stderr and "needs a GPU" not in r.stderr
    if _no_gpu():
        r = _run(["--steps", "1", "--warmup", "0", "--allow-options"], {var: val})
        assert r.returncode != 0 and "refusing" not in r.stderr and "needs a GPU" in r.stderr


def test_bench_reads_its_kernel_configuration_from_the_library():
    """`config.kernel_options` of the line: run-time options, the debug switches of csrc/mlp_fused.hip and the stamps hook, read from
    the library's host state (no device call).  Defaults -> "default": true; an option set at run time -> refused unless allowed."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from tianshou_marl_amd import ops

    cfg = bench.kernel_configuration()
    assert cfg["default"] is True and cfg["stamps_armed"] is False and cfg["update_variant"] == 0 and cfg["slab_store"] == 0
    assert set(ops.KERNEL_OPTIONS) <= set(cfg)

    class A:
        allow_options = False

    out = {}
    bench.guard_configuration(A, out)
    assert out["config"]["kernel_options"]["default"] is True and "diagnostic" not in out
    with ops.kernel_override(generic_kernels=1):
        with pytest.raises(SystemExit):
            bench.guard_configuration(A, {})
        A.allow_options = True
        out = {}
        bench.guard_configuration(A, out)
        assert out["diagnostic"] is True and out["config"]["kernel_options"]["generic_kernels"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("workload,p2p", [("c2", "0"), ("tag", "0"), ("c2", "1"), ("tag", "1"), ("c2", "auto")])
def test_bench_gpus_2_runs_two_ranks_and_counts_them(workload, p2p):
    """The driver's command shape with N = 2 and nothing around it: two ranks on cuda:0 over gloo; p2p = 1: the gradient
    sums on the one-shot peer-memory all-reduce (csrc/p2p.hip) instead of the process group's; auto (the default, variable
    unset): the first-use handshake decides -- it passes here."""
    extra = ["--no-cpu-baseline", "--no-c3-grid"] if workload == "c2" else ["--workload", "tag", "--tag-envs", "64"]
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "2", *extra],
             {"TSM_SHARE_GPU": "1", "TSM_DIST_BACKEND": "gloo", **({} if p2p == "auto" else {"TSM_P2P_ALLREDUCE": p2p})})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line on stdout, nothing else
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["group_ranks"] == 2 and out["ranks_backend"] == "gloo"
    assert out["replicas_identical"] is True
    assert out["steps"] == 3 and out["value"] > 0
    assert out["config"]["gradient_all_reduce"].startswith("process group (gloo)" if p2p == "0" else "peer memory (one-shot")
