"""ctypes binding of the C ABI in include/salp_vec.h (libsalp_hip.so, built in-tree by
`__graft_entry__.build()` / `csrc/build.py`).

There is no CPU fallback: if the shared library is missing or no HIP device is usable the
constructors raise (`SalpError`), they never route to another implementation.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

from .config import CConfig, SalpSnakeConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsalp_hip.so")

SALP_DEVICE_PTRS = 1
INFO_COLS = 3
# snapshot rows (include/salp_vec.h)
F_X, F_Y, F_VX, F_VY, F_THETA, F_OMEGA, F_NOZZLE, F_WATER, F_ELLIPSE_A, F_ELLIPSE_B, F_FOOD0 = range(11)
(I_PHASE, I_TIMER, I_EXHALE_DUR, I_SHAPE_HOLD, I_STEPS_SINCE_FOOD, I_FOOD_COLLECTED, I_RNG_COUNTER,
 I_EPISODE_LENGTH, I_COUNT) = range(9)

EXPORTS = (
    "salp_last_error", "salp_abi_version", "salp_device_count", "salp_config_default",
    "salp_vec_create", "salp_vec_destroy", "salp_vec_num_envs", "salp_vec_obs_dim", "salp_vec_act_dim",
    "salp_vec_num_food", "salp_vec_device", "salp_vec_reset", "salp_vec_step", "salp_vec_rollout",
    "salp_vec_observe", "salp_vec_get_state", "salp_vec_set_state", "salp_vec_get_stats",
    "salp_vec_clear_stats", "salp_vec_global_step", "salp_vec_set_base_num_food", "salp_vec_base_num_food",
    "salp_vec_reseed", "salp_vec_last_launch", "salp_vec_last_kernel_resources",
)


class SalpError(RuntimeError):
    pass


class CStats(ctypes.Structure):
    _fields_ = [
        ("env_steps", ctypes.c_int64), ("episodes", ctypes.c_int64), ("terminated", ctypes.c_int64),
        ("truncated", ctypes.c_int64), ("collisions", ctypes.c_int64), ("food_collected", ctypes.c_int64),
        ("episode_length_sum", ctypes.c_int64), ("reward_sum", ctypes.c_double),
        ("episode_return_sum", ctypes.c_double),
    ]


_lib = None


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """Loads libsalp_hip.so and declares every prototype.  Raises SalpError if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SALP_HIP_LIBRARY", LIB_PATH)
    # If PyTorch is installed, let it load ITS HIP runtime first: torch ships its own libamdhip64 and
    # cannot initialise once the system copy (which this library would pull in) is already resident,
    # whereas this library runs on either copy.
    if os.environ.get("SALP_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.isfile(p):
        raise SalpError(
            f"HIP library not found at {p}; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    L = ctypes.CDLL(p)
    vp, i64, u64, i32, u32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint32
    L.salp_last_error.restype = ctypes.c_char_p
    L.salp_last_error.argtypes = []
    L.salp_abi_version.argtypes = []
    L.salp_device_count.argtypes = []
    L.salp_config_default.argtypes = [ctypes.POINTER(CConfig)]
    L.salp_vec_create.argtypes = [ctypes.POINTER(CConfig), i64, ctypes.c_int, u64, i64, ctypes.POINTER(vp)]
    L.salp_vec_destroy.argtypes = [vp]
    L.salp_vec_destroy.restype = None
    L.salp_vec_num_envs.argtypes = [vp]
    L.salp_vec_num_envs.restype = i64
    for f in ("salp_vec_obs_dim", "salp_vec_act_dim", "salp_vec_num_food", "salp_vec_device"):
        getattr(L, f).argtypes = [vp]
    L.salp_vec_reset.argtypes = [vp, vp, vp, u32, vp]
    L.salp_vec_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u32, vp]
    L.salp_vec_rollout.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, u32, vp]
    L.salp_vec_observe.argtypes = [vp, vp, u32, vp]
    L.salp_vec_get_state.argtypes = [vp, vp, vp, u32, vp]
    L.salp_vec_set_state.argtypes = [vp, vp, vp, u32, vp]
    L.salp_vec_get_stats.argtypes = [vp, ctypes.POINTER(CStats)]
    L.salp_vec_clear_stats.argtypes = [vp]
    if path is None or hasattr(L, "salp_vec_last_launch"):
        L.salp_vec_last_launch.argtypes = [vp, ctypes.POINTER(ctypes.c_int64)]
    if path is None or hasattr(L, "salp_vec_last_kernel_resources"):
        L.salp_vec_last_kernel_resources.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    if path is None or hasattr(L, "salp_vec_reseed"):   # (an explicit path may be an older A/B variant, profiles/ab_bench.py)
        L.salp_vec_reseed.argtypes = [vp, u64, vp, u32, vp]
    L.salp_vec_global_step.argtypes = [vp]
    L.salp_vec_global_step.restype = i64
    L.salp_vec_set_base_num_food.argtypes = [vp, i32]
    L.salp_vec_base_num_food.argtypes = [vp]
    L.salp_vec_base_num_food.restype = i32
    if path is None:
        _lib = L
    return L


def check(lib, rc: int, what: str):
    if rc != 0:
        msg = lib.salp_last_error()
        raise SalpError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


class SalpLib:
    """One handle of the C ABI (one GPU).  Pointers are raw integers (device or host addresses)."""

    def __init__(self, cfg: SalpSnakeConfig, n_envs: int, device_id: int = 0, seed: int = 0,
                 env_index_base: int = 0):
        self.lib = load_library()
        self.cfg = cfg
        self._c = cfg.to_c()
        self._h = ctypes.c_void_p()
        rc = self.lib.salp_vec_create(ctypes.byref(self._c), int(n_envs), int(device_id),
                                      int(seed) & 0xFFFFFFFFFFFFFFFF, int(env_index_base),
                                      ctypes.byref(self._h))
        check(self.lib, rc, "salp_vec_create")
        self.n_envs = int(self.lib.salp_vec_num_envs(self._h))
        self.obs_dim = int(self.lib.salp_vec_obs_dim(self._h))
        self.act_dim = int(self.lib.salp_vec_act_dim(self._h))
        self.num_food = int(self.lib.salp_vec_num_food(self._h))
        self.device_id = int(self.lib.salp_vec_device(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.salp_vec_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _ptr(x):
        if x is None:
            return None
        if isinstance(x, int):
            return ctypes.c_void_p(x)
        if hasattr(x, "data_ptr"):          # torch tensor
            return ctypes.c_void_p(x.data_ptr())
        return x.ctypes.data_as(ctypes.c_void_p)  # numpy

    def reset(self, mask, obs, flags, stream=0):
        check(self.lib, self.lib.salp_vec_reset(self._h, self._ptr(mask), self._ptr(obs), flags,
                                                ctypes.c_void_p(stream)), "salp_vec_reset")

    def step(self, act, obs, reward, term, trunc, final_obs, info, flags, stream=0):
        check(self.lib, self.lib.salp_vec_step(self._h, self._ptr(act), self._ptr(obs), self._ptr(reward),
                                               self._ptr(term), self._ptr(trunc), self._ptr(final_obs),
                                               self._ptr(info), flags, ctypes.c_void_p(stream)),
              "salp_vec_step")

    def rollout(self, act, horizon, obs, reward, term, trunc, final_obs, act_out, flags, stream=0):
        check(self.lib, self.lib.salp_vec_rollout(self._h, self._ptr(act), int(horizon), self._ptr(obs),
                                                  self._ptr(reward), self._ptr(term), self._ptr(trunc),
                                                  self._ptr(final_obs), self._ptr(act_out), flags,
                                                  ctypes.c_void_p(stream)), "salp_vec_rollout")

    def reseed(self, seed, obs, flags, stream=0):
        """New draw streams keyed by `seed`, every env reset from draw counter 0; no reallocation (capture-safe)."""
        check(self.lib, self.lib.salp_vec_reseed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, self._ptr(obs), flags,
                                                 ctypes.c_void_p(stream)), "salp_vec_reseed")

    def observe(self, obs, flags, stream=0):
        check(self.lib, self.lib.salp_vec_observe(self._h, self._ptr(obs), flags, ctypes.c_void_p(stream)),
              "salp_vec_observe")

    def get_state(self, f64, i32, flags, stream=0):
        check(self.lib, self.lib.salp_vec_get_state(self._h, self._ptr(f64), self._ptr(i32), flags,
                                                    ctypes.c_void_p(stream)), "salp_vec_get_state")

    def set_state(self, f64, i32, flags, stream=0):
        check(self.lib, self.lib.salp_vec_set_state(self._h, self._ptr(f64), self._ptr(i32), flags,
                                                    ctypes.c_void_p(stream)), "salp_vec_set_state")

    def stats(self) -> dict:
        s = CStats()
        check(self.lib, self.lib.salp_vec_get_stats(self._h, ctypes.byref(s)), "salp_vec_get_stats")
        return {k: getattr(s, k) for k, _ in CStats._fields_}

    def last_launch(self) -> dict:
        """The kernel instantiation of the most recent step / rollout call (salp_vec_last_launch).  `full_signature`: 1 = the four
        main outputs only, 2 = the four plus final_obs / info, 0 = some main output absent."""
        a = (ctypes.c_int64 * 8)()
        check(self.lib, self.lib.salp_vec_last_launch(self._h, a), "salp_vec_last_launch")
        keys = ("food_slots", "observed_capacity", "literal_constants", "forced", "full_signature", "actions_in_kernel",
                "envs_unpredicated", "envs_predicated")
        return dict(zip(keys, (int(v) for v in a)))

    def last_kernel_resources(self) -> dict:
        """Registers, LDS, scratch and resident workgroups per CU of the most recent call's kernel (salp_vec_last_kernel_resources)."""
        a = (ctypes.c_int32 * 4)()
        check(self.lib, self.lib.salp_vec_last_kernel_resources(self._h, a), "salp_vec_last_kernel_resources")
        return dict(zip(("vgprs", "lds_bytes", "scratch_bytes", "workgroups_per_cu"), (int(v) for v in a)))

    def clear_stats(self):
        check(self.lib, self.lib.salp_vec_clear_stats(self._h), "salp_vec_clear_stats")

    @property
    def base_num_food(self) -> int:
        return int(self.lib.salp_vec_base_num_food(self._h))

    def set_base_num_food(self, k: int):
        check(self.lib, self.lib.salp_vec_set_base_num_food(self._h, int(k)), "salp_vec_set_base_num_food")

    @property
    def global_step(self) -> int:
        return int(self.lib.salp_vec_global_step(self._h))
