"""Environment factory (reference offpolicy_rnn/env_utils/make_env.py:41-72).

The simulator zoo of the reference (`envs/`) is out of scope.  Names of the form
`synthetic-o<obs>-a<act>-T<len>` (or `-d<n>-` for n discrete actions) build the Gaussian environment used by the benchmark / tests; anything else is
handed to `gym.make` when gym is installed and fails loudly otherwise."""
import re

import numpy as np


class Box:
    def __init__(self, low, high, shape):
        self.low = np.full(shape, low, dtype=np.float32)
        self.high = np.full(shape, high, dtype=np.float32)
        self.shape = tuple(shape)
        self._rs = np.random.RandomState(0)

    def seed(self, s):
        self._rs = np.random.RandomState(s)

    def sample(self):
        return self._rs.uniform(-1, 1, self.shape)


class Discrete:
    def __init__(self, n):
        self.n, self.shape = n, ()
        self._rs = np.random.RandomState(0)

    def seed(self, s):
        self._rs = np.random.RandomState(s)

    def sample(self):
        return int(self._rs.randint(self.n))


class SyntheticEnv:
    """i.i.d. Gaussian observations / rewards, fixed horizon: the synthetic workload of BASELINE.json configs 2-4.
    `discrete=True`: `act_dim` discrete actions instead of a Box."""

    def __init__(self, obs_dim, act_dim, horizon, seed=0, discrete=False):
        self.observation_space = Box(-np.inf, np.inf, (obs_dim,))
        self.action_space = Discrete(act_dim) if discrete else Box(-1.0, 1.0, (act_dim,))
        self.horizon, self.t = horizon, 0
        self._rs = np.random.RandomState(seed)

    def seed(self, s):
        self._rs = np.random.RandomState(s)

    def reset(self, *a):
        self.t = 0
        return self._rs.randn(self.observation_space.shape[0])

    def step(self, action):
        self.t += 1
        return self._rs.randn(self.observation_space.shape[0]), float(self._rs.randn()), self.t >= self.horizon, {}


_SYN = re.compile(r'^synthetic-o(\d+)-([ad])(\d+)-T(\d+)$')          # a<n>: Box(n) actions, d<n>: n discrete actions


def make_env(env_name: str, seed: int) -> dict:
    m = _SYN.match(env_name)
    if m:
        obs, act, T = int(m.group(1)), int(m.group(3)), int(m.group(4))
        disc = m.group(2) == 'd'
        return dict(train_env=SyntheticEnv(obs, act, T, seed, disc), eval_env=SyntheticEnv(obs, act, T, seed + 1, disc), train_tasks=[],
                    eval_tasks=[None], max_rollouts_per_task=1, max_trajectory_len=T, obs_dim=obs, act_dim=act,
                    act_continuous=not disc, seed=seed, multiagent=False)
    try:
        import gym
    except ImportError as e:
        raise ImportError(f'environment {env_name!r} needs `gym` and the reference env zoo, which are outside this build; '
                          f'use synthetic-o<obs>-a<act>-T<len>') from e
    env, eval_env = gym.make(env_name), gym.make(env_name)
    T = getattr(env, '_max_episode_steps', 1000)
    cont = hasattr(env.action_space, 'low')
    return dict(train_env=env, eval_env=eval_env, train_tasks=[], eval_tasks=[None], max_rollouts_per_task=1,
                max_trajectory_len=T, obs_dim=env.observation_space.shape[0],
                act_dim=env.action_space.shape[0] if cont else env.action_space.n, act_continuous=cont, seed=seed, multiagent=False)
