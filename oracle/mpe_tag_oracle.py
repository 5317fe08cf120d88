"""TEST INFRASTRUCTURE ONLY -- numpy float64 restatement of PettingZoo-MPE `simple_tag` (one world).

Checker for tianshou_marl_amd/csrc/mpe_tag.hip; never imported by the product.  Restates the published MPE
specification (pettingzoo 1.24.2 `mpe/simple_tag` + `_mpe_utils/core.py`, pinned in the reference's poetry.lock but
absent from /root/reference and not installed here): PARITY WITH PETTINGZOO IS UNPINNED -- kernel and oracle are
checked against each other only.

World: entities = agents (adversaries first, then good agents) followed by obstacles.  Per step: action force
accel * u, soft contact force between every collidable pair (a < b in entity order), velocity damping 0.25, speed
clamp per agent, explicit Euler with dt 0.1.  Rewards on the new state: good agent -10 per touching adversary minus
the boundary penalty; each adversary +10 per (good, adversary) contact.  Observations zero-padded to a common width.
"""
from __future__ import annotations

import numpy as np


class SimpleTagWorld:
    dt, damping, contact_force, contact_margin = 0.1, 0.25, 100.0, 1e-3

    def __init__(self, n_adv: int = 3, n_good: int = 1, n_obst: int = 2, max_cycles: int = 25, seed: int = 0) -> None:
        self.n_adv, self.n_good, self.n_obst, self.max_cycles = n_adv, n_good, n_obst, max_cycles
        self.NA = n_adv + n_good
        self.size = np.array([0.075] * n_adv + [0.05] * n_good + [0.2] * n_obst)
        self.accel = np.array([3.0] * n_adv + [4.0] * n_good)
        self.vmax = np.array([1.0] * n_adv + [1.3] * n_good)
        self.obs_dim = 4 + 2 * n_obst + 2 * (self.NA - 1) + 2 * n_good
        self.rng = np.random.default_rng(seed)
        self.reset()

    def reset(self):
        self.apos = self.rng.uniform(-1, 1, (self.NA, 2))
        self.avel = np.zeros((self.NA, 2))
        self.lpos = self.rng.uniform(-0.9, 0.9, (self.n_obst, 2))
        self.steps = 0
        return self.observe()

    def set_state(self, apos, avel, lpos, steps=0) -> None:
        self.apos, self.avel = np.array(apos, np.float64), np.array(avel, np.float64)
        self.lpos, self.steps = np.array(lpos, np.float64).reshape(self.n_obst, 2), int(steps)

    def observe(self) -> np.ndarray:
        obs = np.zeros((self.NA, self.obs_dim))
        for i in range(self.NA):
            parts = [self.avel[i], self.apos[i]]
            parts += [self.lpos[l] - self.apos[i] for l in range(self.n_obst)]
            parts += [self.apos[j] - self.apos[i] for j in range(self.NA) if j != i]
            parts += [self.avel[j] for j in range(self.n_adv, self.NA) if j != i]
            v = np.concatenate(parts)
            obs[i, :len(v)] = v
        return obs

    @staticmethod
    def _bound(x: float) -> float:
        if x < 0.9:
            return 0.0
        if x < 1.0:
            return (x - 0.9) * 10
        return min(np.exp(2 * x - 2), 10)

    def step(self, act):
        NA, NE = self.NA, self.NA + self.n_obst
        pos = np.concatenate([self.apos, self.lpos]) if self.n_obst else self.apos.copy()
        force = np.zeros((NA, 2))
        for i, a in enumerate(np.asarray(act).reshape(-1)):
            u = np.zeros(2)
            if a == 1: u[0] = -1.0
            if a == 2: u[0] = +1.0
            if a == 3: u[1] = -1.0
            if a == 4: u[1] = +1.0
            force[i] = u * self.accel[i]
        for a in range(NA):
            for b in range(a + 1, NE):
                delta = pos[a] - pos[b]
                dist = np.sqrt(np.sum(delta * delta))
                dmin = self.size[a] + self.size[b]
                k = self.contact_margin
                pen = np.logaddexp(0, -(dist - dmin) / k) * k
                f = self.contact_force * delta / dist * pen
                force[a] += f
                if b < NA:
                    force[b] -= f
        for i in range(NA):
            v = self.avel[i] * (1 - self.damping) + force[i] * self.dt
            sp = np.sqrt(v[0] ** 2 + v[1] ** 2)
            if sp > self.vmax[i]:
                v = v / sp * self.vmax[i]
            self.avel[i] = v
            self.apos[i] = self.apos[i] + v * self.dt
        rew = np.zeros(NA)
        adv_rew = 0.0
        for g in range(self.n_adv, NA):
            r = 0.0
            for a in range(self.n_adv):
                if np.sqrt(np.sum((self.apos[a] - self.apos[g]) ** 2)) < self.size[a] + self.size[g]:
                    r -= 10
                    adv_rew += 10
            r -= self._bound(abs(self.apos[g, 0])) + self._bound(abs(self.apos[g, 1]))
            rew[g] = r
        rew[:self.n_adv] = adv_rew
        self.steps += 1
        trunc = self.steps >= self.max_cycles
        return self.observe(), rew, np.zeros(NA, bool), np.full(NA, trunc)
