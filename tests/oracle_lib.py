"""ctypes binding of oracle/libsalp_oracle.so — TEST INFRASTRUCTURE (see oracle/salp_oracle.c).

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libsalp_oracle.so")

F_X, F_Y, F_VX, F_VY, F_THETA, F_OMEGA, F_NOZZLE, F_WATER, F_ELLIPSE_A, F_ELLIPSE_B, F_FOOD0 = range(11)
I_PHASE, I_TIMER, I_EXHALE_DUR, I_SHAPE_HOLD, I_STEPS_SINCE_FOOD, I_FOOD_COLLECTED, I_RNG_COUNTER, \
    I_EPISODE_LENGTH, I_COUNT = range(9)
INFO_COLS = 3


def build_oracle(force: bool = False) -> str:
    src = os.path.join(_ROOT, "oracle", "salp_oracle.c")
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(_ROOT, "oracle"), "-s"], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = ctypes.CDLL(_SO)
        vp, i64, u64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int32
        L.salp_oracle_create.argtypes = [vp, i64, u64, i64, ctypes.POINTER(vp)]
        L.salp_oracle_destroy.argtypes = [vp]
        L.salp_oracle_destroy.restype = None
        L.salp_oracle_obs_dim.argtypes = [vp]
        L.salp_oracle_act_dim.argtypes = [vp]
        L.salp_oracle_reset.argtypes = [vp, vp, vp]
        L.salp_oracle_observe.argtypes = [vp, vp]
        L.salp_oracle_step.argtypes = [vp] * 9
        L.salp_oracle_rollout.argtypes = [vp, vp, i32] + [vp] * 8
        L.salp_oracle_rollout_f64.argtypes = [vp, vp, i32, vp, vp, vp, vp]
        L.salp_oracle_get_state.argtypes = [vp, vp, vp]
        L.salp_oracle_set_state.argtypes = [vp, vp, vp]
        L.salp_oracle_set_threads.argtypes = [ctypes.c_int]
        L.salp_oracle_set_threads.restype = None
        L.salp_oracle_philox4x32_10.argtypes = [vp, vp, vp]
        L.salp_oracle_philox4x32_10.restype = None
        L.salp_oracle_global_step.argtypes = [vp]
        L.salp_oracle_global_step.restype = i64
        L.salp_oracle_set_base_num_food.argtypes = [vp, ctypes.c_int]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def philox(counter, key):
    c = np.asarray(counter, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().salp_oracle_philox4x32_10(_p(c), _p(k), _p(out))
    return tuple(int(v) for v in out)


class OracleVec:
    """N reference-faithful CPU envs (fp64, libm), with the HIP library's call shapes."""

    def __init__(self, cfg, n_envs: int, seed: int = 0, env_index_base: int = 0, threads: int = 1):
        self.cfg = cfg
        self.n = int(n_envs)
        self.obs_dim = cfg.obs_dim
        self.act_dim = cfg.act_dim
        self.F = cfg.num_food_items
        self._c = cfg.to_c()
        self._h = ctypes.c_void_p()
        lib().salp_oracle_set_threads(int(threads))
        rc = lib().salp_oracle_create(ctypes.byref(self._c), self.n, seed, env_index_base,
                                      ctypes.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"salp_oracle_create failed: {rc}")

    def set_base_num_food(self, k):
        if lib().salp_oracle_set_base_num_food(self._h, int(k)) != 0:
            raise ValueError(f"base_num_food_items {k} out of range")

    def close(self):
        if self._h:
            lib().salp_oracle_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask=None):
        obs = np.empty((self.n, self.obs_dim), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().salp_oracle_reset(self._h, _p(m), _p(obs))
        return obs

    def observe(self):
        obs = np.empty((self.n, self.obs_dim), np.float32)
        lib().salp_oracle_observe(self._h, _p(obs))
        return obs

    def step(self, act, want_final=False):
        act = np.ascontiguousarray(act, dtype=np.float32).reshape(self.n, self.act_dim)
        out = self.rollout(act[None], want_final=want_final)
        return {k: (v[0] if v is not None else None) for k, v in out.items()}

    def rollout(self, act=None, horizon=None, want_final=False, light=False):
        if act is not None:
            act = np.ascontiguousarray(act, dtype=np.float32).reshape(-1, self.n, self.act_dim)
            horizon = act.shape[0]
        H, n = int(horizon), self.n
        if light:  # timing leg: only the last-step outputs are kept by the caller
            obs = reward = r64 = term = trunc = info = None
        else:
            obs = np.empty((H, n, self.obs_dim), np.float32)
            reward = np.empty((H, n), np.float32)
            r64 = np.empty((H, n), np.float64)
            term = np.empty((H, n), np.uint8)
            trunc = np.empty((H, n), np.uint8)
            info = np.empty((H, n, INFO_COLS), np.int32)
        fin = np.full((H, n, self.obs_dim), np.nan, np.float32) if want_final else None
        aout = np.empty((H, n, self.act_dim), np.float32) if (act is None and not light) else None
        rc = lib().salp_oracle_rollout(self._h, _p(act), H, _p(obs), _p(reward), _p(r64), _p(term),
                                       _p(trunc), _p(fin), _p(info), _p(aout))
        if rc != 0:
            raise RuntimeError(f"salp_oracle_rollout failed: {rc}")
        return dict(obs=obs, reward=reward, reward64=r64, terminated=term, truncated=trunc,
                    final_obs=fin, info=info, actions=aout)

    def rollout_f64(self, act64):
        """fp64 actions [H, n, act_dim] (test-only entry point, see oracle/salp_oracle.h)."""
        a = np.ascontiguousarray(act64, dtype=np.float64).reshape(-1, self.n, self.act_dim)
        H = a.shape[0]
        obs = np.empty((H, self.n, self.obs_dim), np.float32)
        r64 = np.empty((H, self.n), np.float64)
        term = np.empty((H, self.n), np.uint8)
        trunc = np.empty((H, self.n), np.uint8)
        rc = lib().salp_oracle_rollout_f64(self._h, _p(a), H, _p(obs), _p(r64), _p(term), _p(trunc))
        if rc != 0:
            raise RuntimeError(f"salp_oracle_rollout_f64 failed: {rc}")
        return dict(obs=obs, reward64=r64, terminated=term, truncated=trunc)

    def get_state(self):
        f64 = np.empty((F_FOOD0 + 2 * self.F, self.n), np.float64)
        i32 = np.empty((I_COUNT, self.n), np.int32)
        lib().salp_oracle_get_state(self._h, _p(f64), _p(i32))
        return f64, i32

    def set_state(self, f64=None, i32=None):
        f = None if f64 is None else np.ascontiguousarray(f64, dtype=np.float64)
        i = None if i32 is None else np.ascontiguousarray(i32, dtype=np.int32)
        lib().salp_oracle_set_state(self._h, _p(f), _p(i))
