"""Batched SALP swimmer simulator for AMD MI355X (gfx950).

Drop-in for the `SalpSnakeEnv.step()` hot path of SungRoboticsGroup/UNDERWATER-SWIMMER_RL
(src/salp/environments/salp_snake_env.py over scripts/utilities/salp_robot.py): a fused
HIP kernel behind the C ABI of include/salp_vec.h, and a Gymnasium-VectorEnv-style host shim.
"""
from .config import SalpSnakeConfig, load_env_config, PRESETS  # noqa: F401

__all__ = ["SalpSnakeConfig", "load_env_config", "PRESETS"]


def __getattr__(name):  # lazy: the pieces below need torch / the HIP library
    if name in ("SalpVectorEnv", "SalpSB3VecEnv"):
        from . import vector_env
        return getattr(vector_env, name)
    if name in ("ShardedSalpVectorEnv",):
        from . import sharded
        return getattr(sharded, name)
    if name in ("SalpLib", "load_library", "SalpError"):
        from . import _capi
        return getattr(_capi, name)
    raise AttributeError(name)
