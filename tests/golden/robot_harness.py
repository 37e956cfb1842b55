"""Runs the reference's HEAD simulator (src/salp/environments/robot.py + salp_robot_env.py) in this
container — TEST TOOLING ONLY, like ref_harness.py.  The env module is loaded with the in-memory
`gymnasium` / `pygame` stand-ins; its `np.random.uniform` (target placement, salp_robot_env.py:247-250) is
redirected to the build's Philox stream (stream id 16, one block per episode: x from u53(w0,w1), y from
u53(w2,w3)).  Actions are handed over as float64 arrays (see include/salp_robot.h)."""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np

import ref_harness as rh

ENV_DIR = os.path.join(rh.REFERENCE_ROOT, "src/salp/environments")


def available() -> bool:
    return os.path.isfile(os.path.join(ENV_DIR, "robot.py")) and os.path.isfile(os.path.join(ENV_DIR, "salp_robot_env.py"))


class _UniformProxy:
    def __init__(self, owner):
        self._o = owner

    def uniform(self, lo, hi):
        return self._o.next_uniform(lo, hi)


class _NpProxy:
    def __init__(self):
        self.random = _UniformProxy(self)
        self.stream = None

    def next_uniform(self, lo, hi):
        return self.stream.uniform(lo, hi)

    def __getattr__(self, name):
        return getattr(np, name)


class TargetStream:
    def __init__(self, seed, env_index):
        self.key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        self.env = (env_index & 0xFFFFFFFF, (env_index >> 32) & 0xFFFFFFFF)
        self.episode = 0
        self._pending = None

    def uniform(self, lo, hi):
        if self._pending is None:
            w = rh.philox4x32_10((self.env[0], self.env[1], self.episode, 16), self.key)
            self.episode += 1
            u, self._pending = rh.u53(w[0], w[1]), rh.u53(w[2], w[3])
        else:
            u, self._pending = self._pending, None
        return lo + (hi - lo) * u


_LOADED = None


def load():
    global _LOADED
    if _LOADED is None:
        rh._install_standins()
        if ENV_DIR not in sys.path:
            sys.path.insert(0, ENV_DIR)
        import robot as robot_mod  # the reference's robot.py (numpy only)
        spec = importlib.util.spec_from_file_location("salp_robot_env_head", os.path.join(ENV_DIR, "salp_robot_env.py"))
        env_mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(env_mod)
        proxy = _NpProxy()
        env_mod.np = proxy
        _LOADED = (robot_mod, env_mod, proxy)
    return _LOADED


class ReferenceRobotEnv:
    """make_env() of train_robot.py:10-22 with the target stream of (seed, env_index)."""

    def __init__(self, seed, env_index):
        robot_mod, env_mod, self._proxy = load()
        nozzle = robot_mod.Nozzle(length1=0.05, length2=0.05, length3=0.05, area=0.00016, mass=1.0)
        robot = robot_mod.Robot(dry_mass=1.0, init_length=0.3, init_width=0.15, max_contraction=0.06, nozzle=nozzle)
        robot.nozzle.set_angles(angle1=0.0, angle2=0.0)
        robot.set_environment(density=1000)
        self.stream = TargetStream(seed, env_index)
        self._proxy.stream = TargetStream(0xBAD, 0xBAD)      # the constructor's own reset() burns a throw-away stream
        self.env = env_mod.SalpRobotEnv(render_mode=None, robot=robot)

    def reset(self):
        self._proxy.stream = self.stream
        obs, _ = self.env.reset()
        return obs

    def step(self, action):
        self._proxy.stream = self.stream
        return self.env.step(np.asarray(action, dtype=np.float64))

    def state(self):
        r = self.env.robot
        return dict(pos=r.position.copy(), vel=r.velocity.copy(), euler=r.euler_angle.copy(), omega=r.angular_velocity.copy(),
                    vel_world=r.velocity_world.copy(), prev_I=np.diag(r.prev_I).copy(), target=np.asarray(self.env.target_point).copy(),
                    prev_dist=float(self.env.prev_dist), volume=float(r.volume), angle1=float(r.nozzle.angle1),
                    angle2=float(r.nozzle.angle2), time=float(r.time), cycle=int(r.cycle))
