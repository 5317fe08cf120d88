"""Host CPU budget helpers (tianshou_marl_amd/utils/host.py): no reference counterpart -- they exist because torch sizes its
intra-op pool from the machine, not from the container's CPU quota, and a throttled launch thread is a 30-60 ms hole on the
device timeline (profiles/r03_host_stall.txt)."""
import warnings

import pytest

torch = pytest.importorskip("torch")

from tianshou_marl_amd.utils import host  # noqa: E402


def test_cpu_budget_is_positive_and_within_the_machine():
    import os

    b = host.cpu_budget()
    assert 1 <= b <= (os.cpu_count() or 1)


def test_limit_host_threads_sets_the_pool_and_defaults_to_half_the_budget(monkeypatch):
    before = torch.get_num_threads()
    try:
        assert host.limit_host_threads(3) == 3 and torch.get_num_threads() == 3
        monkeypatch.setattr(host, "cpu_budget", lambda: 16)
        assert host.limit_host_threads() == 8 and torch.get_num_threads() == 8
        monkeypatch.setattr(host, "cpu_budget", lambda: 1)
        assert host.limit_host_threads() == 1
        monkeypatch.setattr(host, "cpu_budget", lambda: 512)
        assert host.limit_host_threads() == 16  # never more than 16: the launch path has nothing to parallelise
    finally:
        torch.set_num_threads(before)


def test_oversubscription_is_reported_once(monkeypatch):
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(4)
        monkeypatch.setattr(host, "_warned", False)
        monkeypatch.setattr(host, "cpu_budget", lambda: 2)
        with pytest.warns(RuntimeWarning, match="CFS quota"):
            host.warn_if_oversubscribed()
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            host.warn_if_oversubscribed()  # once per process
        monkeypatch.setattr(host, "_warned", False)
        monkeypatch.setattr(host, "cpu_budget", lambda: 64)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            host.warn_if_oversubscribed()  # within budget: silent
    finally:
        torch.set_num_threads(before)
