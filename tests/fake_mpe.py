"""A stand-in for ``harl.envs.pettingzoo_mpe.pettingzoo_mpe_env`` (the real one needs pettingzoo + supersuit, which are not
installed here): a single-environment class with the env contract the reference's vector wrappers drive
(harl/envs/env_wrappers.py:166-211, README.md "Application to new environments"): ``reset() -> (obs, state, avail)``,
``step(actions) -> (obs, state, rewards, dones, infos, avail)``, ``n_agents``, ``observation_space``,
``share_observation_space``, ``action_space``, ``seed()``, ``close()``.

Shapes follow MPE simple_spread_v2: 3 agents, obs 18, state 54, Box(5) when ``continuous_actions`` else Discrete(5); every
``max_cycles`` (25) steps all agents are done and the info carries ``bad_transition`` (pettingzoo_mpe_env.py:50-53).
The dynamics are a toy cooperative task (agents are rewarded for matching a smooth target of the state).
The space classes live in tests/fake_env.py: they travel through the subprocess workers' pipes, so they must be importable."""
import numpy as np

from tests.fake_env import Box, Discrete


class PettingZooMPEEnv:
    def __init__(self, args):
        self.n_agents, self.od, self.sd, self.ad = 3, 18, 54, 5
        self.discrete = not bool(args.get("continuous_actions", False))
        self.max_cycles = int(args.get("max_cycles", 25))
        self.observation_space = [Box((self.od,))] * self.n_agents
        self.share_observation_space = [Box((self.sd,))] * self.n_agents
        self.action_space = [Discrete(self.ad) if self.discrete else Box((self.ad,))] * self.n_agents
        self.rng = np.random.default_rng(0)
        self.W = np.random.default_rng(123).standard_normal((self.n_agents, self.ad, self.sd)).astype(np.float32) / np.sqrt(self.sd)
        self.P = np.random.default_rng(124).standard_normal((self.n_agents, self.od, self.sd)).astype(np.float32) / np.sqrt(self.sd)
        self.t = 0
        self.s = np.zeros(self.sd, dtype=np.float32)

    def seed(self, seed):
        self.rng = np.random.default_rng(seed)

    def _out(self):
        obs = [self.P[a] @ self.s for a in range(self.n_agents)]
        state = [self.s.copy() for _ in range(self.n_agents)]
        avail = [[1] * self.ad for _ in range(self.n_agents)] if self.discrete else None
        return obs, state, avail

    def reset(self):
        self.t = 0
        self.s = self.rng.standard_normal(self.sd).astype(np.float32)
        return self._out()

    def step(self, actions):
        actions = np.asarray(actions)
        tgt = self.W @ self.s  # [A, ad]
        if self.discrete:
            r = float(np.mean(actions.reshape(self.n_agents).astype(np.int64) == tgt.argmax(-1)))
        else:
            r = -float(np.mean(np.sum((actions.reshape(self.n_agents, self.ad) - 0.6 * np.tanh(tgt)) ** 2, -1)))
        self.t += 1
        done = self.t >= self.max_cycles
        self.s = (0.9 * self.s + 0.3 * self.rng.standard_normal(self.sd)).astype(np.float32)
        obs, state, avail = self._out()
        infos = [{"bad_transition": True} if done else {} for _ in range(self.n_agents)]
        return obs, state, [[r]] * self.n_agents, [done] * self.n_agents, infos, avail

    def close(self):
        pass
