"""Optimizer / LR-scheduler factories with the reference's names and arguments.

Mirror of /root/reference/tianshou/algorithm/optim.py:14-138.  In the reference a factory builds a
`torch.optim.Adam` over `module.parameters()` plus (optionally) a `LambdaLR`, and `Algorithm._update` calls
`lr_scheduler.step()` once per update (algorithm_base.py:626-627).  Here the optimizer IS the HIP Adam kernel
(csrc/adam.hip) on the algorithm's flat parameter vector, so a factory only carries hyper-parameters, and a scheduler
moves the algorithm's learning rate -- which lives in HBM (`algo._lr_dev`, read by `tsm_adam_step(lr_dev=...)`), so the
captured update graph follows the schedule without being re-captured.
"""
from __future__ import annotations

import numpy as np


class LambdaLR:
    """torch.optim.lr_scheduler.LambdaLR on an object with a settable `lr`: lr = base_lr * lr_lambda(last_epoch).
    As in torch, construction performs the initial step (last_epoch = 0)."""

    def __init__(self, target, lr_lambda) -> None:
        self.target, self.lr_lambda = target, lr_lambda
        self.base_lr = float(target.lr)
        self.last_epoch = -1
        self.step()

    def step(self) -> None:
        self.last_epoch += 1
        self.target.lr = self.base_lr * float(self.lr_lambda(self.last_epoch))

    def get_last_lr(self) -> list[float]:
        return [float(self.target.lr)]

    def state_dict(self) -> dict:
        return {"base_lrs": [self.base_lr], "last_epoch": self.last_epoch, "_last_lr": self.get_last_lr()}

    def load_state_dict(self, sd: dict) -> None:
        self.base_lr, self.last_epoch = float(sd["base_lrs"][0]), int(sd["last_epoch"])
        self.target.lr = float(sd["_last_lr"][0]) if "_last_lr" in sd else self.base_lr * float(self.lr_lambda(self.last_epoch))


class LRSchedulerFactory:
    """optim.py:14-19."""

    def create_scheduler(self, optim):
        raise NotImplementedError


class LRSchedulerFactoryLinear(LRSchedulerFactory):
    """optim.py:22-46: the learning rate decays linearly towards zero over the updates of a training run."""

    def __init__(self, max_epochs: int, epoch_num_steps: int, collection_step_num_env_steps: int) -> None:
        self.num_epochs = max_epochs
        self.epoch_num_steps = epoch_num_steps
        self.collection_step_num_env_steps = collection_step_num_env_steps

    def create_scheduler(self, optim) -> LambdaLR:
        max_update_num = np.ceil(self.epoch_num_steps / self.collection_step_num_env_steps) * self.num_epochs
        return LambdaLR(optim, lr_lambda=lambda epoch: 1.0 - epoch / max_update_num)


class OptimizerFactory:
    """optim.py:49-70."""

    def __init__(self) -> None:
        self.lr_scheduler_factory: LRSchedulerFactory | None = None

    def with_lr_scheduler_factory(self, lr_scheduler_factory: LRSchedulerFactory) -> "OptimizerFactory":
        self.lr_scheduler_factory = lr_scheduler_factory
        return self


class AdamOptimizerFactory(OptimizerFactory):
    """optim.py:91-111 (same defaults)."""

    def __init__(self, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-08,
                 weight_decay: float = 0) -> None:
        super().__init__()
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay

    def adam_kwargs(self) -> dict:
        return dict(lr=self.lr, betas=tuple(self.betas), adam_eps=self.eps, weight_decay=self.weight_decay)
