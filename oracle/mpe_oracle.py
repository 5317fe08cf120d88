"""TEST INFRASTRUCTURE ONLY -- numpy (f64) restatement of PettingZoo-MPE `simple_spread` dynamics.

pettingzoo 1.24.2 (poetry.lock pin of the reference) is NOT in /root/reference and not installed here,
so this restates the published MPE specification (dt 0.1, damping 0.25, contact_force 100,
contact_margin 1e-3, agent size 0.15, accel 5, Discrete(5) actions, local_ratio reward mix, truncation
at max_cycles) and is **parity-unpinned** against pettingzoo itself.  It checks the HIP env kernel
(csrc/mpe.hip) and serves bench.py's cpu_baseline leg as the per-env Python-loop environment that the
reference would drive through EnhancedPettingZooEnv + DummyVectorEnv
(/root/reference/tianshou/env/enhanced_pettingzoo_env.py:175-222, venvs.py:281-287).
"""
from __future__ import annotations

import numpy as np

DT, DAMPING, CONTACT_FORCE, CONTACT_MARGIN = 0.1, 0.25, 100.0, 1e-3
AGENT_SIZE, ACCEL = 0.15, 5.0


class SimpleSpreadWorld:
    """One world (the object a PettingZoo ParallelEnv wraps)."""

    def __init__(self, n_agent: int = 3, max_cycles: int = 25, local_ratio: float = 0.5, seed: int = 0) -> None:
        self.N, self.max_cycles, self.local_ratio = n_agent, max_cycles, local_ratio
        self.rng = np.random.default_rng(seed)
        self.agents = [f"agent_{i}" for i in range(n_agent)]
        self.reset()

    def reset(self):
        self.apos = self.rng.uniform(-1, 1, (self.N, 2))
        self.avel = np.zeros((self.N, 2))
        self.lpos = self.rng.uniform(-1, 1, (self.N, 2))
        self.steps = 0
        return self.observe()

    def set_state(self, apos, avel, lpos, steps=0) -> None:
        self.apos, self.avel, self.lpos = np.array(apos, np.float64), np.array(avel, np.float64), np.array(lpos, np.float64)
        self.steps = int(steps)

    def observe(self) -> np.ndarray:
        obs = np.zeros((self.N, 6 * self.N))
        for i in range(self.N):
            o = [self.avel[i], self.apos[i]]
            o += [self.lpos[l] - self.apos[i] for l in range(self.N)]
            o += [self.apos[j] - self.apos[i] for j in range(self.N) if j != i]
            o += [np.zeros(2) for j in range(self.N) if j != i]
            obs[i] = np.concatenate(o)
        return obs

    def step(self, act):
        N = self.N
        f = np.zeros((N, 2))
        for i, a in enumerate(act):
            if a == 1: f[i, 0] = -1.0
            if a == 2: f[i, 0] = +1.0
            if a == 3: f[i, 1] = -1.0
            if a == 4: f[i, 1] = +1.0
        f *= ACCEL
        for i in range(N):
            for j in range(i + 1, N):
                delta = self.apos[i] - self.apos[j]
                dist = np.sqrt(np.sum(delta**2))
                pen = np.logaddexp(0, -(dist - 2 * AGENT_SIZE) / CONTACT_MARGIN) * CONTACT_MARGIN
                force = CONTACT_FORCE * delta / dist * pen
                f[i] += force
                f[j] -= force
        self.avel = self.avel * (1 - DAMPING) + f * DT
        self.apos = self.apos + self.avel * DT
        glob = 0.0
        for l in range(N):
            glob -= min(np.sqrt(np.sum((self.apos[i] - self.lpos[l]) ** 2)) for i in range(N))
        rew = np.zeros(N)
        for i in range(N):
            local = 0.0
            for j in range(N):
                if j != i and np.sqrt(np.sum((self.apos[i] - self.apos[j]) ** 2)) < 2 * AGENT_SIZE:
                    local -= 1.0
            rew[i] = glob * (1 - self.local_ratio) + local * self.local_ratio
        self.steps += 1
        trunc = self.steps >= self.max_cycles
        return self.observe(), rew, np.zeros(N, bool), np.full(N, trunc)
