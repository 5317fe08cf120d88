"""The on-policy outer loop with the reference's parameter object, so `algorithm.run_training(OnPolicyTrainerParams(...))`
of a reference script works on the device engine.

Mirror of the PATH of /root/reference/tianshou/trainer/trainer.py that the hot loop lives in:
    OnPolicyTrainerParams                          :79-303 (same field names, defaults, validation)
    Trainer.run / execute_epoch / _test_step       :557-757
    OnlineTrainer._training_step / _collect_training_data / _test_in_train   :878-983
    OnPolicyTrainer._update_step                   :1079-1109
Loggers, progress bars, resume-from-log and the moving-average loss bookkeeping are out of scope; a
`logger` object, when given, only receives `log_train_data / log_test_data / log_update_data / log_info_data` calls if
it has those methods.  One training step = collect -> (optional early-stop test) -> update -> reset_buffer(keep
statistics): exactly the two device-side calls `bench.py` times.
"""
from __future__ import annotations

import time
from dataclasses import asdict, dataclass, field
from typing import Any, Callable

import numpy as np

from .algorithm.ppo import policy_within_training_step
from .data.stats import CollectStats, SequenceSummaryStats, TrainingStats


@dataclass(kw_only=True)
class OnPolicyTrainerParams:
    max_epochs: int = 100
    epoch_num_steps: int = 30000
    test_collector: Any = None
    test_step_num_episodes: int = 1
    train_fn: Callable[[int, int], None] | None = None
    test_fn: Callable[[int, int | None], None] | None = None
    stop_fn: Callable[[float], bool] | None = None
    compute_score_fn: Callable[[CollectStats], float] | None = None
    save_best_fn: Callable[[Any], None] | None = None
    save_checkpoint_fn: Callable[[int, int, int], str] | None = None
    resume_from_log: bool = False
    multi_agent_return_reduction: Callable[[np.ndarray], np.ndarray] | None = None
    logger: Any = None
    verbose: bool = True
    show_progress: bool = True
    train_collector: Any = None
    collection_step_num_env_steps: int | None = 2048
    collection_step_num_episodes: int | None = None
    test_in_train: bool = False
    batch_size: int | None = 64
    update_step_num_repetitions: int = 1

    def __post_init__(self) -> None:  # trainer.py:207-228, 275-282
        if self.train_collector is None:
            raise TypeError("OnPolicyTrainerParams: train_collector is required")
        if self.resume_from_log:
            raise ValueError("resume_from_log is not supported (loggers are outside the hot path)")
        if self.test_collector is None:
            for name in ("stop_fn", "test_fn", "save_best_fn"):
                if getattr(self, name) is not None:
                    raise ValueError(f"{name} is set while test steps are disabled (test_collector is None)")
        elif self.test_step_num_episodes < 1:
            raise ValueError("test_step_num_episodes must be positive if test steps are enabled (test_collector not None)")
        if (self.collection_step_num_env_steps is None) == (self.collection_step_num_episodes is None):
            raise ValueError("Exactly one of {collection_step_num_env_steps, collection_step_num_episodes} must be set")
        if self.test_in_train and (self.test_collector is None or self.stop_fn is None):
            raise ValueError("test_in_train requires test_collector and stop_fn to be set")


@dataclass(kw_only=True)
class TimingStats:
    total_time: float = 0.0
    train_time: float = 0.0
    train_time_collect: float = 0.0
    train_time_update: float = 0.0
    test_time: float = 0.0
    update_speed: float = 0.0


@dataclass(kw_only=True)
class InfoStats:
    update_step: int
    best_score: float
    best_reward: float
    best_reward_std: float
    train_step: int
    train_episode: int
    test_step: int
    test_episode: int
    timing: TimingStats = field(default_factory=TimingStats)


@dataclass(kw_only=True)
class EpochStats:
    epoch: int
    train_collect_stat: Any
    test_collect_stat: Any
    training_stat: Any
    info_stat: InfoStats


def _plain(stats) -> dict:
    """asdict() for the eager dataclass; the lazy collect statistics are resolved first."""
    if hasattr(stats, "resolve"):
        stats.resolve()
    try:
        return asdict(stats)
    except TypeError:
        return {k: v for k, v in vars(stats).items() if not k.startswith("_")}


class OnPolicyTrainer:
    def __init__(self, algorithm, params: OnPolicyTrainerParams) -> None:
        self.algorithm, self.params = algorithm, params
        self._epoch = self._env_step = self._current_update_step = self._env_episode = 0
        self._policy_update_time = 0.0
        self._best_score = self._best_reward = self._best_reward_std = 0.0
        self._best_epoch = 0
        self._stop_fn_flag = False
        self._start_time = time.time()
        # A training loop is where a frozen launch thread costs device time: cap torch's intra-op pool to the CPU quota of the
        # process (utils/host.py has the measurement; TSM_LIMIT_HOST_THREADS=0 leaves the pool alone)
        import os

        if os.environ.get("TSM_LIMIT_HOST_THREADS", "1") != "0":
            from .utils.host import cpu_budget, limit_host_threads

            import torch

            if torch.get_num_threads() > cpu_budget():
                limit_host_threads()

    # ---- helpers ---------------------------------------------------------------------------------------------------
    def _log(self, method: str, *args) -> None:
        fn = getattr(self.params.logger, method, None)
        if fn is not None:
            fn(*args)

    def _score(self, stats: CollectStats) -> float:
        if self.params.compute_score_fn is not None:
            return self.params.compute_score_fn(stats)
        return stats.returns_stat.mean  # trainer.py:394-403

    def _reduce_returns(self, stats) -> None:
        red = self.params.multi_agent_return_reduction
        if red is not None and stats.n_collected_episodes > 0:  # trainer.py:176-185, 941-945
            rew = red(stats.returns)
            stats.returns = rew
            stats.returns_stat = SequenceSummaryStats.from_sequence(rew)

    def _should_stop(self, *, score: float | None = None, collect_stats=None) -> bool:
        if self.params.stop_fn is None:
            return False
        if score is None:
            if collect_stats.n_collected_episodes == 0:
                return False
            score = self._score(collect_stats)
        return bool(self.params.stop_fn(score))

    # ---- life cycle (trainer.py:418-456, 836-851, 736-757) ------------------------------------------------------
    def reset(self, reset_collectors: bool = True, reset_collector_buffers: bool = False) -> None:
        self._epoch = self._env_step = self._current_update_step = self._env_episode = 0
        self._start_time = time.time()
        if reset_collectors:
            self.params.train_collector.reset(reset_buffer=reset_collector_buffers)
            if self.params.test_collector is not None:
                self.params.test_collector.reset(reset_buffer=reset_collector_buffers)
        if self.params.test_collector is not None:
            self._test_step(force_update_best=True, prefix="Initial test step")
        self._stop_fn_flag = False

    def run(self, reset_collectors: bool = True, reset_collector_buffers: bool = False) -> InfoStats:
        self.reset(reset_collectors=reset_collectors, reset_collector_buffers=reset_collector_buffers)
        while self._epoch < self.params.max_epochs and not self._stop_fn_flag:
            self.execute_epoch()
        return self._create_info_stats()

    def execute_epoch(self) -> EpochStats:
        self._epoch += 1
        done_in_epoch, collect_stats, training_stats = 0, None, None
        while done_in_epoch < self.params.epoch_num_steps and not self._stop_fn_flag:
            self._current_update_step += 1
            collect_stats, training_stats, self._stop_fn_flag = self._training_step()
            done_in_epoch += collect_stats.n_collected_steps
            self._env_step += collect_stats.n_collected_steps
            self._log("log_train_data", _plain(collect_stats), self._env_step)
        test_stats = None
        if not self._stop_fn_flag:
            if self.params.save_checkpoint_fn is not None:
                self.params.save_checkpoint_fn(self._epoch, self._env_step, self._current_update_step)
            if self.params.test_collector is not None:
                test_stats, self._stop_fn_flag = self._test_step()
        info = self._create_info_stats()
        self._log("log_info_data", asdict(info), self._epoch)
        return EpochStats(epoch=self._epoch, train_collect_stat=collect_stats, test_collect_stat=test_stats,
                          training_stat=training_stats, info_stat=info)

    # ---- one training step (trainer.py:878-951, 1079-1109) --------------------------------------------------------
    def _training_step(self):
        with policy_within_training_step(self.algorithm.policy):
            collect_stats = self._collect_training_data()
            stop = self._test_in_train(collect_stats) if self.params.test_in_train else False
            training_stats = None if stop else self._update_step(collect_stats)
        return collect_stats, training_stats, stop

    def _collect_training_data(self):
        p = self.params
        if p.train_fn:
            p.train_fn(self._epoch, self._env_step)
        stats = p.train_collector.collect(n_step=p.collection_step_num_env_steps, n_episode=p.collection_step_num_episodes)
        if p.train_collector.buffer.hasnull():
            from ._abi import MalformedBufferError

            raise MalformedBufferError(f"Encountered NaNs in buffer after {self._env_step} steps.")
        self._reduce_returns(stats)
        self._env_episode += stats.n_collected_episodes
        return stats

    def _test_in_train(self, train_stats) -> bool:
        if train_stats.n_collected_episodes > 0 and self._should_stop(collect_stats=train_stats):
            with policy_within_training_step(self.algorithm.policy, enabled=False):
                _, stop = self._test_step(prefix=f"Test step triggered by train stats (env_step={self._env_step})")
            return stop
        return False

    def _update_step(self, collect_stats=None) -> TrainingStats:
        p = self.params
        stat = self.algorithm.update(buffer=p.train_collector.buffer, batch_size=p.batch_size,
                                     repeat=p.update_step_num_repetitions)
        self._policy_update_time += stat.train_time
        # the update has consumed the rollout: erase the rows but keep the running episode return / length of episodes
        # cut by the collection boundary (trainer.py:1095-1104)
        p.train_collector.reset_buffer(keep_statistics=True)
        self._log("log_update_data", stat.get_loss_stats_dict(), self._current_update_step)
        return stat

    # ---- test step (trainer.py:640-706) -----------------------------------------------------------------------------
    def _test_step(self, force_update_best: bool = False, prefix: str | None = None):
        p = self.params
        p.test_collector.reset(reset_stats=False)
        if p.test_fn:
            p.test_fn(self._epoch, self._env_step)
        stat = p.test_collector.collect(n_episode=p.test_step_num_episodes)
        self._reduce_returns(stat)
        self._log("log_test_data", _plain(stat), self._env_step)
        rew, rew_std = stat.returns_stat.mean, stat.returns_stat.std
        score = self._score(stat)
        if score > self._best_score or force_update_best:
            self._best_score, self._best_epoch = score, self._epoch
            self._best_reward, self._best_reward_std = float(rew), rew_std
            if p.save_best_fn:
                p.save_best_fn(self.algorithm)
        if p.verbose:
            print(f"{prefix or f'Epoch #{self._epoch}'}: test_reward: {rew:.6f} ± {rew_std:.6f}, best_reward: "
                  f"{self._best_reward:.6f} ± {self._best_reward_std:.6f} in #{self._best_epoch}", flush=True)
        return stat, self._should_stop(score=self._best_score)

    def _create_info_stats(self) -> InfoStats:
        tr, te = self.params.train_collector, self.params.test_collector
        duration = max(0.0, time.time() - self._start_time)
        test_time = te.collect_time if te is not None else 0.0
        timing = TimingStats(total_time=duration, train_time=duration - test_time, train_time_collect=tr.collect_time,
                             train_time_update=self._policy_update_time, test_time=test_time,
                             update_speed=tr.collect_step / max(duration - test_time, 1e-9))
        return InfoStats(update_step=self._current_update_step, best_score=self._best_score, best_reward=self._best_reward,
                         best_reward_std=self._best_reward_std, train_step=tr.collect_step, train_episode=tr.collect_episode,
                         test_step=te.collect_step if te is not None else 0,
                         test_episode=te.collect_episode if te is not None else 0, timing=timing)
