"""Batched HEAD simulator: the reference's `SalpRobotEnv(robot=Robot(...))`
(src/salp/environments/salp_robot_env.py over robot.py) for N robots on one MI355X.

    env = SalpRobotVectorEnv(num_envs=65536, device="cuda:0", seed=0)
    obs, _ = env.reset()
    obs, reward, terminated, truncated, info = env.step(actions)      # actions [N, 3] in [0,1]x[0,1]x[-1,1]

One `step` is one whole breathing cycle of every robot (Robot.set_control + step_through_cycle,
robot.py:335-358, 422-445): `info["inner_steps"]` is the number of Euler steps each robot took.
Same conventions as SalpVectorEnv: device tensors, same-step autoreset with `final_observation`,
no CPU fallback."""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import numpy as np

from . import _capi
from .spaces import Box, batch_space

(R_POS, R_VEL, R_EULER, R_OMEGA, R_VEL_WORLD, R_PREV_I, R_TARGET, R_PREV_DIST, R_VOLUME, R_ANGLE1, R_ANGLE2, R_TIME,
 R_CYCLE, R_RNG, R_COUNT) = 0, 3, 6, 9, 12, 15, 18, 20, 21, 22, 23, 24, 25, 26, 27

ROBOT_EXPORTS = ("salp_robot_last_error", "salp_robot_config_default", "salp_robot_vec_create", "salp_robot_vec_destroy",
                 "salp_robot_vec_num_envs", "salp_robot_vec_reset", "salp_robot_vec_step", "salp_robot_vec_get_state")


class CRobotConfig(ctypes.Structure):
    """salp_robot_config_t (include/salp_robot.h)."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("tank_margin", ctypes.c_double), ("dry_mass", ctypes.c_double), ("init_length", ctypes.c_double),
                ("init_width", ctypes.c_double), ("max_contraction", ctypes.c_double), ("density", ctypes.c_double),
                ("dt", ctypes.c_double), ("drag_coefficient_min", ctypes.c_double), ("drag_coefficient_max", ctypes.c_double),
                ("nozzle_length1", ctypes.c_double), ("nozzle_length2", ctypes.c_double), ("nozzle_length3", ctypes.c_double),
                ("nozzle_area", ctypes.c_double), ("nozzle_mass", ctypes.c_double), ("nozzle_gamma", ctypes.c_double),
                ("max_cycles", ctypes.c_int32), ("reserved0", ctypes.c_int32)]


def _lib():
    L = _capi.load_library()
    if not getattr(L, "_salp_robot_ready", False):
        vp, i64, u64, u32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32
        L.salp_robot_last_error.restype = ctypes.c_char_p
        L.salp_robot_config_default.argtypes = [ctypes.POINTER(CRobotConfig)]
        L.salp_robot_vec_create.argtypes = [ctypes.POINTER(CRobotConfig), i64, ctypes.c_int, u64, i64, ctypes.POINTER(vp)]
        L.salp_robot_vec_destroy.argtypes = [vp]
        L.salp_robot_vec_destroy.restype = None
        L.salp_robot_vec_num_envs.argtypes = [vp]
        L.salp_robot_vec_num_envs.restype = i64
        L.salp_robot_vec_reset.argtypes = [vp, vp, vp, u32, vp]
        L.salp_robot_vec_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u32, vp]
        L.salp_robot_vec_get_state.argtypes = [vp, vp, u32, vp]
        L._salp_robot_ready = True
    return L


def _check(L, rc, what):
    if rc != 0:
        raise _capi.SalpError(f"{what} failed ({rc}): {L.salp_robot_last_error().decode()}")


class SalpRobotVectorEnv:
    def __init__(self, num_envs: int = 4096, device="cuda:0", seed: int = 0, env_index_base: int = 0,
                 output: str = "torch", **robot_params):
        self.L = _lib()
        self.num_envs = int(num_envs)
        cfg = CRobotConfig()
        _check(self.L, self.L.salp_robot_config_default(ctypes.byref(cfg)), "salp_robot_config_default")
        for k, v in robot_params.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown robot parameter {k}")
            setattr(cfg, k, v)
        self.cfg = cfg
        self._dev_index = int(str(device).split(":")[1]) if isinstance(device, str) and ":" in device else int(device) if not isinstance(device, str) else 0
        self._h = ctypes.c_void_p()
        _check(self.L, self.L.salp_robot_vec_create(ctypes.byref(cfg), self.num_envs, self._dev_index, int(seed),
                                                    int(env_index_base), ctypes.byref(self._h)), "salp_robot_vec_create")
        self.obs_dim, self.act_dim = 6, 3
        # salp_robot_env.py:48-69
        self.single_action_space = Box(low=np.array([0.0, 0.0, -1.0]), high=np.array([1.0, 1.0, 1.0]), dtype=np.float32)
        self.single_observation_space = Box(low=np.full(6, -np.inf), high=np.full(6, np.inf), dtype=np.float32)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self._torch = None
        if output == "torch":
            import torch
            if not torch.cuda.is_available():
                raise _capi.SalpError("output='torch' needs a ROCm GPU visible to PyTorch")
            self._torch, self.device = torch, torch.device("cuda", self._dev_index)
        n = self.num_envs
        self._obs, self._fin = self._new((n, 6), np.float32), self._new((n, 6), np.float32)
        self._rew, self._term, self._trunc = self._new((n,), np.float32), self._new((n,), np.uint8), self._new((n,), np.uint8)
        self._inner = self._new((n,), np.int32)

    def _new(self, shape, dtype):
        if self._torch is not None:
            td = {np.float32: self._torch.float32, np.uint8: self._torch.uint8, np.int32: self._torch.int32}[dtype]
            return self._torch.empty(shape, dtype=td, device=self.device)
        return np.empty(shape, dtype)

    @staticmethod
    def _p(x):
        if x is None:
            return None
        return ctypes.c_void_p(x.data_ptr()) if hasattr(x, "data_ptr") else x.ctypes.data_as(ctypes.c_void_p)

    @property
    def _flags(self):
        return 1 if self._torch is not None else 0

    @property
    def _stream(self):
        return ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream) if self._torch is not None else None

    def reset(self, *, seed: Optional[int] = None, options=None, mask=None):
        m = None
        if mask is not None:
            m = (self._torch.as_tensor(mask).to(self.device, self._torch.uint8).contiguous() if self._torch is not None
                 else np.ascontiguousarray(mask, np.uint8))
        _check(self.L, self.L.salp_robot_vec_reset(self._h, self._p(m), self._p(self._obs), self._flags, self._stream), "reset")
        return self._obs, {}

    def observe(self):
        zero = (self._torch.zeros(self.num_envs, dtype=self._torch.uint8, device=self.device) if self._torch is not None
                else np.zeros(self.num_envs, np.uint8))
        _check(self.L, self.L.salp_robot_vec_reset(self._h, self._p(zero), self._p(self._obs), self._flags, self._stream), "observe")
        return self._obs

    def step(self, actions):
        if self._torch is not None:
            t = self._torch
            a = actions if isinstance(actions, t.Tensor) else t.as_tensor(np.asarray(actions, np.float32))
            a = a.to(self.device, t.float32).reshape(self.num_envs, 3).contiguous()
        else:
            a = np.ascontiguousarray(np.asarray(actions, np.float32).reshape(self.num_envs, 3))
        _check(self.L, self.L.salp_robot_vec_step(self._h, self._p(a), self._p(self._obs), self._p(self._rew), self._p(self._term),
                                                   self._p(self._trunc), self._p(self._fin), self._p(self._inner), self._flags,
                                                   self._stream), "step")
        if self._torch is not None:
            term, trunc = self._term.view(self._torch.bool), self._trunc.view(self._torch.bool)
        else:
            term, trunc = self._term.view(np.bool_), self._trunc.view(np.bool_)
        info = {"inner_steps": self._inner, "final_observation": self._fin, "_final_observation": term | trunc}
        return self._obs, self._rew, term, trunc, info

    def get_state(self) -> np.ndarray:
        """fp64 [27, N] snapshot, rows R_* (include/salp_robot.h)."""
        s = np.empty((R_COUNT, self.num_envs), np.float64)
        if self._torch is not None:
            self._torch.cuda.current_stream(self.device).synchronize()
        _check(self.L, self.L.salp_robot_vec_get_state(self._h, s.ctypes.data_as(ctypes.c_void_p), 0, None), "get_state")
        return s

    def close(self):
        if self._h:
            self.L.salp_robot_vec_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
