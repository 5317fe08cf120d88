"""The readiness script of the AsyncCollector fixture (tests/golden/async_collector.npz): ONE rule for the reference run
(make_fixtures.py::make_async_collector, through the worker class's `wait`) and for the build's replay (through the vector env's
`ready_selector`), so that both see the same interleaving.  Test infrastructure; no reference code."""


def scripted_ready(n_waiting: int, wait_num: int, call: int) -> list[int]:
    """Which of the waiting envs (positions in waiting order) return from async step() call number `call`: min(wait_num,
    n_waiting) of them, starting at a position that rotates with the call count."""
    k = min(wait_num, n_waiting)
    start = call % n_waiting
    return [(start + t) % n_waiting for t in range(k)]
