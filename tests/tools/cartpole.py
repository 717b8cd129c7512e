"""CartPole-v1 as a host environment for BASELINE configs[0] (the reference's plumbing case: CPU actors, CPU
learner).  The physics is the published cart-pole of Barto, Sutton & Anderson (1983) in the parametrisation
every gym release uses for CartPole-v1 (gravity 9.8, cart 1.0 kg, pole 0.1 kg, half-length 0.5 m, force 10 N,
Euler steps of 0.02 s; an episode ends beyond +-2.4 m or +-12 degrees, or after 500 steps; reward 1 per step;
reset draws the four state variables from U(-0.05, 0.05)).  gym is not installed in this image and is not
part of the reference tree; this is a restatement of the equations, test infrastructure only."""
import math

import numpy as np


class CartPole(object):
    GRAVITY, M_CART, M_POLE, HALF_LEN, FORCE, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    X_LIMIT, THETA_LIMIT, MAX_STEPS = 2.4, 12 * 2 * math.pi / 360, 500
    obs_dim, act_dim = 4, 2

    def __init__(self, seed=0):
        self.rng = np.random.default_rng(seed)
        self.state, self.steps = None, 0

    def reset(self):
        self.state = self.rng.uniform(-0.05, 0.05, 4)
        self.steps = 0
        return self.state.astype(np.float32)

    def step(self, action):
        x, x_dot, th, th_dot = self.state
        f = self.FORCE if int(action) == 1 else -self.FORCE
        total_m, pm_l = self.M_CART + self.M_POLE, self.M_POLE * self.HALF_LEN
        c, s = math.cos(th), math.sin(th)
        tmp = (f + pm_l * th_dot * th_dot * s) / total_m
        th_acc = (self.GRAVITY * s - c * tmp) / (self.HALF_LEN * (4.0 / 3.0 - self.M_POLE * c * c / total_m))
        x_acc = tmp - pm_l * th_acc * c / total_m
        self.state = np.array([x + self.TAU * x_dot, x_dot + self.TAU * x_acc,
                               th + self.TAU * th_dot, th_dot + self.TAU * th_acc])
        self.steps += 1
        fell = abs(self.state[0]) > self.X_LIMIT or abs(self.state[2]) > self.THETA_LIMIT
        done = bool(fell or self.steps >= self.MAX_STEPS)
        return self.state.astype(np.float32), 1.0, done, {}


class MonitoredCartPole(CartPole):
    """the MonitorEnv bookkeeping the actors' get_metrics reads (atari_wrappers.py:44-100): episode returns /
    lengths closed since the last poll"""

    def __init__(self, seed=0):
        super(MonitoredCartPole, self).__init__(seed)
        self._ret, self._closed = 0.0, []

    def reset(self):
        self._ret = 0.0
        return super(MonitoredCartPole, self).reset()

    def step(self, action):
        o, r, d, i = super(MonitoredCartPole, self).step(action)
        self._ret += r
        if d:
            self._closed.append((self._ret, self.steps))
        return o, r, d, i

    def next_episode_results(self):
        out, self._closed = self._closed, []
        return out


class HostVectorEnv(object):
    """parl/env/vector_env.py:26-63 for host envs: step every env, a done env is reset and its RESET observation
    is what the step returns"""

    def __init__(self, envs):
        self.envs, self.envs_num = envs, len(envs)

    def reset(self):
        return [e.reset() for e in self.envs]

    def step(self, actions):
        obs, rews, dones, infos = [], [], [], []
        for e, a in zip(self.envs, actions):
            o, r, d, i = e.step(a)
            if d:
                o = e.reset()
            obs.append(o)
            rews.append(r)
            dones.append(d)
            infos.append(i)
        return obs, rews, dones, infos
