"""The schedules of the trainer-trace fixture (tests/golden/trainers.npz): ONE definition for the reference run
(make_fixtures.py::make_trainers) and for the build's replay (tests/test_host_marl.py).  Test infrastructure; no reference code."""
SEED = 1234
SELFPLAY_STEPS, SNAPSHOT_INTERVAL, POOL_SIZE = 48, 3, 4
LEAGUE_STEPS, LEAGUE_AGENTS, GAMES_PER_EVALUATION = 60, 5, 6
SEQUENTIAL_STEPS, SIMULTANEOUS_STEPS = 11, 12


def selfplay_sampling(i: int) -> str:
    return ("uniform", "prioritized", "latest")[(i // 5) % 3]


def selfplay_won(i: int) -> bool:
    return (5 * i + 1) % 3 != 0


def league_matchmaking(i: int) -> str:
    return ("random", "elo", "win_rate")[(i // 7) % 3]


def league_winner_first(i: int) -> bool:
    """Whether the first agent of step i's match wins."""
    return (3 * i + 1) % 4 != 0
