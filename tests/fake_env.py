"""A vectorised toy multi-agent environment with the reference's ``ShareVecEnv`` surface (harl/envs/env_wrappers.py:
``reset() -> (obs, share_obs, available_actions)``, ``step(actions) -> (obs, share_obs, rewards, dones, infos,
available_actions)``, NumPy in / NumPy out), used to drive ``OnPolicyHARunner.run()`` end to end on the GPU box.

State s in R^ds evolves as an AR(1) process per thread.  Box: agent a is rewarded for matching a fixed smooth target
``0.6 tanh(W_a s)``; Discrete: for picking ``argmax(W_a s)`` among the available actions.  The team reward is shared
(cooperative, like MPE simple_spread).  Every ``horizon`` steps all agents are done and the info carries
``bad_transition`` (time-limit truncation, pettingzoo_mpe_env.py:50-53); agent 1 additionally "dies" (done alone) for a
few steps so that active_masks are exercised.
"""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, shape):
        self.shape = shape


class Discrete:
    def __init__(self, n):
        self.n = n


class FakeVecEnv:
    def __init__(self, n_threads: int, n_agents: int = 3, state_dim: int = 6, act_dim: int = 2, discrete: bool = False,
                 horizon: int = 25, seed: int = 0):
        self.N, self.A, self.ds, self.da, self.discrete, self.horizon = n_threads, n_agents, state_dim, act_dim, discrete, horizon
        self.rng = np.random.default_rng(seed)
        self.W = self.rng.standard_normal((n_agents, act_dim, state_dim)).astype(np.float32) / np.sqrt(state_dim)
        od = state_dim + n_agents
        self.observation_space = [Box((od,))] * n_agents
        self.share_observation_space = [Box((state_dim,))] * n_agents
        self.action_space = [Discrete(act_dim) if discrete else Box((act_dim,))] * n_agents
        self.t = np.zeros(n_threads, dtype=np.int64)
        self.s = np.zeros((n_threads, state_dim), dtype=np.float32)

    def _obs(self):
        eye = np.eye(self.A, dtype=np.float32)
        obs = np.concatenate([np.repeat(self.s[:, None, :], self.A, 1), np.repeat(eye[None], self.N, 0)], -1)
        share = np.repeat(self.s[:, None, :], self.A, 1)
        avail = None
        if self.discrete:  # action (t mod da) is unavailable unless it is the best one
            avail = np.ones((self.N, self.A, self.da), dtype=np.float32)
            best = np.einsum("adk,nk->nad", self.W, self.s).argmax(-1)
            ban = (self.t % self.da)[:, None].repeat(self.A, 1)
            m = ban != best
            n_i, a_i = np.nonzero(m)
            avail[n_i, a_i, ban[n_i, a_i]] = 0.0
        return obs.astype(np.float32), share.astype(np.float32), avail

    def reset(self):
        self.t[:] = 0
        self.s = self.rng.standard_normal((self.N, self.ds)).astype(np.float32)
        obs, share, avail = self._obs()
        return obs, share, (avail if avail is not None else np.array([None] * self.N))

    def step(self, actions):
        actions = np.asarray(actions)
        tgt = np.einsum("adk,nk->nad", self.W, self.s)
        if self.discrete:
            r = (actions[..., 0].astype(np.int64) == tgt.argmax(-1)).astype(np.float32).mean(1)
        else:
            r = -((actions - 0.6 * np.tanh(tgt)) ** 2).sum(-1).mean(1)
        rewards = np.repeat(r[:, None, None], self.A, 1).astype(np.float32)
        self.t += 1
        done_env = self.t >= self.horizon
        dones = np.repeat(done_env[:, None], self.A, 1)
        if self.A > 1:  # agent 1 is "dead" for steps 10..12 of every episode
            dones[:, 1] |= (self.t >= 10) & (self.t <= 12)
        infos = [[{"bad_transition": bool(done_env[n])} for _ in range(self.A)] for n in range(self.N)]
        self.s = (0.9 * self.s + 0.3 * self.rng.standard_normal(self.s.shape)).astype(np.float32)
        if done_env.any():
            k = int(done_env.sum())
            self.s[done_env] = self.rng.standard_normal((k, self.ds)).astype(np.float32)
            self.t[done_env] = 0
        obs, share, avail = self._obs()
        return obs, share, rewards, dones, infos, (avail if avail is not None else np.array([None] * self.N))

    def close(self):
        pass
