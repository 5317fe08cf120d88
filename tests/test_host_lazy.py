"""CPU tests of host-side helpers that carry no arithmetic: the lazily resolved statistics mapping `PPO.learn` returns and
the chained-rows marker of the device buffer's bookkeeping (tianshou_marl_amd/algorithm/ppo.py, data/buffer.py)."""
import json

import pytest

torch = pytest.importorskip("torch")

from tianshou_marl_amd.algorithm.ppo import LazyLosses  # noqa: E402


class _Event:
    def __init__(self):
        self.waits = 0

    def synchronize(self):
        self.waits += 1


def _slot():
    h = torch.tensor([[1.0, 2.0, 3.0, 4.0], [3.0, 4.0, 5.0, 6.0]])
    return dict(h=h, event=_Event(), pending=None)


def test_lazy_losses_resolve_once_on_first_read_and_behave_like_a_dict():
    slot = _slot()
    d = LazyLosses(slot)
    slot["pending"] = d
    assert slot["event"].waits == 0            # nothing read yet: the host has not waited
    assert d["loss"] == 2.0 and slot["event"].waits == 1
    assert d == {"loss": 2.0, "actor_loss": 3.0, "vf_loss": 4.0, "ent_loss": 5.0}
    assert set(d) == {"loss", "actor_loss", "vf_loss", "ent_loss"} and len(d) == 4 and "vf_loss" in d
    assert d.get("nope", 7) == 7 and dict(d)["ent_loss"] == 5.0 and {**d}["actor_loss"] == 3.0
    assert json.loads(json.dumps(d)) == dict(d)  # (resolved by now: the C encoder walks the dict storage itself)
    assert slot["event"].waits == 1 and slot["pending"] is None  # resolved exactly once, slot released


@pytest.mark.parametrize("reader", [lambda d: list(d.items()), lambda d: list(d.values()), lambda d: repr(d), lambda d: d.copy(),
                                    lambda d: json.dumps(d, indent=1), lambda d: dict(d), lambda d: d == {}, lambda d: len(d)])
def test_every_way_of_reading_lazy_losses_waits_for_the_statistics(reader):
    slot = _slot()
    d = LazyLosses(slot)
    reader(d)
    assert slot["event"].waits == 1 and dict.__len__(d) == 4


def test_two_lazy_results_compare_by_value():
    a, b = LazyLosses(_slot()), LazyLosses(_slot())
    assert a == b and not (a != b)
    with pytest.raises(TypeError):
        hash(a)
