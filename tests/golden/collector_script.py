"""The policy and env rules of the Collector fixture (tests/golden/collector.npz): ONE definition for the reference run
(make_fixtures.py::make_collector) and for the build's replay (tests/test_gpu_marl.py), so both sides act alike.  Test
infrastructure; no reference code."""
import numpy as np

SIZES = (2, 3, 4, 5, 3)           # MoveToRight lengths of the five envs
LIMITS = (0, 0, 3, 0, 2)          # step limit (0: none): envs 2 and 4 are TRUNCATED before they can reach the end
PLAN = (("n_step", 10, ""), ("n_step", 7, ""), ("n_episode", 3, ""), ("n_episode", 8, ""), ("n_step", 5, "reset_before_collect"),
        ("n_step", 15, "reset_buffer"), ("n_episode", 5, "reset_stat"), ("n_step", 5, ""), ("n_episode", 1, ""), ("n_step", 20, ""))


def scripted_action(call: int, obs: np.ndarray) -> np.ndarray:
    """Right (1), except left (0) on every fourth (forward call number + position in the call) where the env is not at 0: three
    steps right for every step left, so every episode ends."""
    pos = np.arange(len(obs))
    at = obs.reshape(len(obs), -1)[:, 0]
    return (~(((call + pos) % 4 == 3) & (at >= 1))).astype(np.int64)


def env_step(index: int, steps: int, size: int, limit: int, action: int):
    """-> (new index, new step count, reward, terminated, truncated): reaching `size` terminates with reward size + 1; every
    step to the left costs 0.5; `limit` steps without reaching the end truncate the episode."""
    index = index + 1 if int(action) == 1 else max(0, index - 1)
    steps += 1
    term = index == size
    trunc = (not term) and limit > 0 and steps >= limit
    return index, steps, float(term) * (size + 1) - 0.5 * (int(action) == 0), term, trunc
