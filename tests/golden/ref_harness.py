"""Reference harness: runs the *reference's own* SalpSnakeEnv in this container.

TEST TOOLING ONLY (never imported by the product, bench.py or any `-m gpu` test;
/root/reference does not exist on the GPU box).  It is used by
`tests/golden/gen_golden.py` to emit the committed golden vectors and by the
`not gpu` test `tests/test_oracle_vs_reference.py`, which skips itself when
/root/reference is absent.

What it does:
  * `SalpSnakeEnv` (src/salp/environments/salp_snake_env.py:17) subclasses a parent
    that at HEAD cannot construct it; the parent it was written against survives as
    scripts/utilities/salp_robot.py:15.  We load that file under the module name
    `salp.environments.salp_robot_env` and then load the snake file.  No reference
    file is modified or copied.
  * `gymnasium` / `pygame` are not installed.  The hot path touches only
    `gym.Env.reset(seed=)` and `spaces.Box(...)` (salp_robot.py:29,80-91,96;
    salp_snake_env.py:71-88); pygame only inside render().  Tiny in-memory stand-in
    modules are registered for those two names; they contribute no arithmetic.
  * The reference draws from two global, un-seeded Mersenne-Twister streams:
    `random.uniform/randint` (salp_snake_env.py:101-104,127-130,146,239-242,269-272)
    and `np.random.random()` (salp_robot.py:311).  We swap the module-namespace
    names `snake.random` and `legacy.np` for proxies that hand out the values of the
    build's counter-based stream (Philox4x32-10, key=seed, counter=(env, draw#)) in
    program order, so the reference, the C oracle and the HIP kernel all see the
    same draws.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("SALP_REFERENCE_ROOT", "/root/reference")
LEGACY_PARENT = os.path.join(REFERENCE_ROOT, "scripts/utilities/salp_robot.py")
SNAKE = os.path.join(REFERENCE_ROOT, "src/salp/environments/salp_snake_env.py")


def reference_available() -> bool:
    return os.path.isfile(LEGACY_PARENT) and os.path.isfile(SNAKE)


# --------------------------------------------------------------------------- Philox
_M0, _M1 = 0xD2511F53, 0xCD9E8D57
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., SC'11; Random123).  Pure-Python ints."""
    c0, c1, c2, c3 = (int(c) & _MASK for c in counter)
    k0, k1 = (int(k) & _MASK for k in key)
    for r in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _MASK, p1 & _MASK, ((p0 >> 32) ^ c3 ^ k1) & _MASK, p0 & _MASK
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def u53(hi_word: int, lo_word: int) -> float:
    """Two 32-bit words -> double in [0,1) with 53 random bits (27 + 26)."""
    return ((hi_word >> 5) * 67108864.0 + (lo_word >> 6)) / 9007199254740992.0


class EnvStream:
    """The per-env draw stream shared by reference proxy, C oracle and HIP kernel.

    Block n of env e under seed s is philox4x32_10((e_lo, e_hi, n, 0), (s_lo, s_hi)).
    Every *draw event* consumes one block:
      jitter  (np.random.random)      -> u53(w0, w1)
      food try (uniform x, uniform y) -> x from u53(w0, w1), y from u53(w2, w3)
      randint(1, n)                   -> 1 + ((w0 * n) >> 32)
    """

    def __init__(self, seed: int, env_index: int, counter: int = 0):
        self.key = (seed & _MASK, (seed >> 32) & _MASK)
        self.env = (env_index & _MASK, (env_index >> 32) & _MASK)
        self.counter = counter
        self._pending_y = None
        self.log = []  # (kind, value) for debugging

    def _block(self):
        w = philox4x32_10((self.env[0], self.env[1], self.counter & _MASK, 0), self.key)
        self.counter += 1
        return w

    def jitter(self) -> float:
        assert self._pending_y is None
        w = self._block()
        u = u53(w[0], w[1])
        self.log.append(("jitter", u))
        return u

    def uniform(self, a: float, b: float) -> float:
        if self._pending_y is None:
            w = self._block()
            u = u53(w[0], w[1])
            self._pending_y = u53(w[2], w[3])
        else:
            u = self._pending_y
            self._pending_y = None
        v = a + (b - a) * u  # CPython random.uniform: a + (b-a) * self.random()
        self.log.append(("uniform", v))
        return v

    def randint(self, a: int, b: int) -> int:
        assert self._pending_y is None
        w = self._block()
        n = b - a + 1
        v = a + ((w[0] * n) >> 32)
        self.log.append(("randint", v))
        return v


class _RandomProxy:
    """Stands in for the `random` module inside salp_snake_env's namespace."""

    def __init__(self):
        self.stream: EnvStream | None = None

    def uniform(self, a, b):
        return self.stream.uniform(a, b)

    def randint(self, a, b):
        return self.stream.randint(a, b)


class _NumpyRandomProxy:
    def __init__(self, owner):
        self._owner = owner

    def random(self):
        return self._owner.stream.jitter()


class _NumpyProxy:
    """Stands in for `np` inside salp_robot's namespace: only `.random.random()` is
    redirected; every other attribute is numpy's own."""

    def __init__(self):
        self.stream: EnvStream | None = None
        self.random = _NumpyRandomProxy(self)

    def __getattr__(self, name):
        return getattr(np, name)


# --------------------------------------------------------------------------- stand-ins
def _install_standins():
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        spaces = types.ModuleType("gymnasium.spaces")

        class Env:  # the two methods the hot path touches
            def reset(self, seed=None, options=None):
                return None

            def close(self):
                return None

        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                if shape is None:
                    shape = np.shape(low)
                self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
                self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
                self.shape = tuple(shape)
                self.dtype = np.dtype(dtype)

            def __repr__(self):
                return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

        gym.Env = Env
        spaces.Box = Box
        gym.spaces = spaces
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces
    if "pygame" not in sys.modules:
        sys.modules["pygame"] = types.ModuleType("pygame")


_LOADED = None


def load_reference():
    """Returns (SalpSnakeEnv class, random proxy, numpy proxy)."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not reference_available():
        raise FileNotFoundError(f"reference not found under {REFERENCE_ROOT}")
    _install_standins()
    for name in ("salp", "salp.environments"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []  # namespace-like
            sys.modules[name] = m

    def _load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    legacy = _load("salp.environments.salp_robot_env", LEGACY_PARENT)
    snake = _load("salp.environments.salp_snake_env", SNAKE)
    rproxy = _RandomProxy()
    nproxy = _NumpyProxy()
    snake.random = rproxy
    legacy.np = nproxy
    _LOADED = (snake.SalpSnakeEnv, rproxy, nproxy)
    return _LOADED


class ReferenceEnv:
    """One reference SalpSnakeEnv wired to one EnvStream.

    The constructor burns its draws on a throw-away stream (the reference's __init__
    generates food twice, salp_snake_env.py:62->151 and :90); the env's real stream
    starts at counter 0 with the first explicit reset().
    """

    def __init__(self, seed: int, env_index: int, **params):
        cls, self._rproxy, self._nproxy = load_reference()
        self.stream = EnvStream(seed, env_index)
        self._bind(EnvStream(0xDEAD, 0xBEEF))
        self.env = cls(render_mode=None, **params)
        self._bind(self.stream)

    def _bind(self, stream):
        self._rproxy.stream = stream
        self._nproxy.stream = stream

    def reset(self):
        self._bind(self.stream)
        obs, info = self.env.reset()
        return obs

    def step(self, action):
        self._bind(self.stream)
        return self.env.step(np.asarray(action))

    # state snapshot in the build's SoA vocabulary
    def state(self):
        e = self.env
        phase = {"rest": 0, "inhaling": 1, "exhaling": 2}[e.breathing_phase]
        foods = [(p[0], p[1]) if p is not None else (float("nan"), float("nan")) for p in e.food_positions]
        return dict(
            x=float(e.robot_pos[0]), y=float(e.robot_pos[1]),
            vx=float(e.robot_velocity[0]), vy=float(e.robot_velocity[1]),
            theta=float(e.robot_angle), omega=float(e.robot_angular_velocity),
            nozzle=float(e.nozzle_angle), water=float(e.water_volume),
            ellipse_a=float(e.ellipse_a), ellipse_b=float(e.ellipse_b),
            phase=phase, timer=int(e.breathing_timer),
            exhale_dur=int(getattr(e, "current_exhale_duration", 0)),  # 0 = not yet set (legacy:233 getattr default is only read while exhaling)
            food=np.array(foods, dtype=np.float64).reshape(-1, 2),
            steps_since_food=int(e.steps_since_food), food_collected=int(e.food_collected),
            score=float(e.score), rng_counter=int(self.stream.counter),
        )
