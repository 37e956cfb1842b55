"""ctypes binding of the HEAD-simulator oracle (oracle/salp_robot_oracle.c) — TEST INFRASTRUCTURE."""
import ctypes

import numpy as np

import oracle_lib as ol

R_POS, R_VEL, R_EULER, R_OMEGA, R_VEL_WORLD, R_PREV_I, R_TARGET, R_PREV_DIST, R_VOLUME, R_ANGLE1, R_ANGLE2, R_TIME, \
    R_CYCLE, R_RNG, R_COUNT = 0, 3, 6, 9, 12, 15, 18, 20, 21, 22, 23, 24, 25, 26, 27


class CRobotConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("tank_margin", ctypes.c_double), ("dry_mass", ctypes.c_double), ("init_length", ctypes.c_double),
                ("init_width", ctypes.c_double), ("max_contraction", ctypes.c_double), ("density", ctypes.c_double),
                ("dt", ctypes.c_double), ("drag_coefficient_min", ctypes.c_double), ("drag_coefficient_max", ctypes.c_double),
                ("nozzle_length1", ctypes.c_double), ("nozzle_length2", ctypes.c_double), ("nozzle_length3", ctypes.c_double),
                ("nozzle_area", ctypes.c_double), ("nozzle_mass", ctypes.c_double), ("nozzle_gamma", ctypes.c_double),
                ("max_cycles", ctypes.c_int32), ("reserved0", ctypes.c_int32)]


def default_robot_config() -> CRobotConfig:
    c = CRobotConfig()
    c.struct_size = ctypes.sizeof(CRobotConfig)
    c.width, c.height, c.tank_margin = 900, 700, 50.0
    c.dry_mass, c.init_length, c.init_width, c.max_contraction, c.density, c.dt = 1.0, 0.3, 0.15, 0.06, 1000.0, 0.01
    c.drag_coefficient_min, c.drag_coefficient_max = 0.4, 1.0
    c.nozzle_length1 = c.nozzle_length2 = c.nozzle_length3 = 0.05
    c.nozzle_area, c.nozzle_mass, c.nozzle_gamma, c.max_cycles = 0.00016, 1.0, np.pi / 4, 500
    return c


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class RobotOracleVec:
    def __init__(self, n, seed=0, env_index_base=0, cfg=None):
        L = ol.lib()
        vp = ctypes.c_void_p
        L.salp_robot_oracle_create.argtypes = [vp, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int64, ctypes.POINTER(vp)]
        L.salp_robot_oracle_destroy.argtypes = [vp]
        L.salp_robot_oracle_destroy.restype = None
        L.salp_robot_oracle_reset.argtypes = [vp, vp, vp]
        L.salp_robot_oracle_step.argtypes = [vp] * 8
        L.salp_robot_oracle_get_state.argtypes = [vp, vp]
        self.L, self.n = L, int(n)
        self.cfg = cfg or default_robot_config()
        self.h = vp()
        rc = L.salp_robot_oracle_create(ctypes.byref(self.cfg), self.n, seed, env_index_base, ctypes.byref(self.h))
        assert rc == 0, rc

    def close(self):
        if self.h:
            self.L.salp_robot_oracle_destroy(self.h)
            self.h = ctypes.c_void_p()

    def reset(self, mask=None):
        obs = np.empty((self.n, 6), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.L.salp_robot_oracle_reset(self.h, _p(m), _p(obs))
        return obs

    def step(self, act):
        a = np.ascontiguousarray(act, np.float32).reshape(self.n, 3)
        obs = np.empty((self.n, 6), np.float32)
        fin = np.full((self.n, 6), np.nan, np.float32)
        rew = np.empty(self.n, np.float64)
        term = np.empty(self.n, np.uint8)
        trunc = np.empty(self.n, np.uint8)
        inner = np.empty(self.n, np.int32)
        self.L.salp_robot_oracle_step(self.h, _p(a), _p(obs), _p(rew), _p(term), _p(trunc), _p(fin), _p(inner))
        return dict(obs=obs, reward=rew, terminated=term, truncated=trunc, final_obs=fin, inner_steps=inner)

    def get_state(self):
        s = np.empty((R_COUNT, self.n), np.float64)
        self.L.salp_robot_oracle_get_state(self.h, _p(s))
        return s
