"""pytest configuration: marker registration + shared paths.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol export (runs on CPU).
`-m gpu`      : HIP path vs oracle / fixtures through the C-ABI on a real MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    from tianshou_marl_amd.utils.host import limit_host_threads

    limit_host_threads()  # torch's 128-thread default against a 16-CPU quota freezes the process for tens of ms (utils/host.py)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc  # noqa: PLC0415

    orc.build()
    return orc
