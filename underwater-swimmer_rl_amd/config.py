"""Environment parameters of the batched SALP simulator.

Mirrors the 13 constructor kwargs of the reference's `SalpSnakeEnv.__init__`
(src/salp/environments/salp_snake_env.py:29-33) and the constants its parent sets
(scripts/utilities/salp_robot.py:32-53), in the layout of `salp_config_t`
(include/salp_vec.h).  `load_env_config()` reads the `environment:` block of the
reference's YAML presets (configs/single_food.yaml etc.; train.py:50-52 forwards
`environment.params` as kwargs and leaves width/height at their defaults).
"""
from __future__ import annotations

import ctypes
import dataclasses
import math
import os
from typing import Any, Dict, Optional

MAX_FOOD = 16
MAX_OBSERVED_FOOD = 8


class CConfig(ctypes.Structure):
    """ctypes image of salp_config_t (include/salp_vec.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("width", ctypes.c_int32),
        ("height", ctypes.c_int32),
        ("num_food_items", ctypes.c_int32),
        ("max_observed_food", ctypes.c_int32),
        ("max_steps_without_food", ctypes.c_int32),
        ("forced_breathing", ctypes.c_int32),
        ("random_food_count", ctypes.c_int32),
        ("respawn_food", ctypes.c_int32),
        ("food_reward", ctypes.c_double),
        ("collision_penalty", ctypes.c_double),
        ("time_penalty", ctypes.c_double),
        ("efficiency_bonus", ctypes.c_double),
        ("proximity_reward_weight", ctypes.c_double),
        ("tank_margin", ctypes.c_double),
        ("base_radius", ctypes.c_double),
        ("max_thrust_force", ctypes.c_double),
        ("drag_coefficient", ctypes.c_double),
        ("angular_drag", ctypes.c_double),
        ("max_nozzle_angle", ctypes.c_double),
        ("nozzle_response_rate", ctypes.c_double),
        ("food_radius", ctypes.c_double),
        ("min_food_distance", ctypes.c_double),
        ("inhale_duration", ctypes.c_int32),
        ("exhale_duration", ctypes.c_int32),
        ("rest_duration", ctypes.c_int32),
        ("no_autoreset", ctypes.c_int32),
    ]


@dataclasses.dataclass
class SalpSnakeConfig:
    # --- SalpSnakeEnv.__init__ kwargs (snake:29-33), same names and defaults
    width: int = 800
    height: int = 600
    num_food_items: int = 5
    food_reward: float = 10.0
    collision_penalty: float = -50.0
    time_penalty: float = -0.1
    efficiency_bonus: float = 1.0
    forced_breathing: bool = True
    max_observed_food: int = 3
    random_food_count: bool = False
    respawn_food: bool = True
    proximity_reward_weight: float = 0.0
    max_steps_without_food: int = 1500
    # --- parent constants (legacy:32-53) and snake:53-54
    tank_margin: float = 50.0
    base_radius: float = 30.0
    max_thrust_force: float = 100.0
    drag_coefficient: float = 0.98
    angular_drag: float = 0.95
    max_nozzle_angle: float = math.pi / 3
    nozzle_response_rate: float = 0.05
    food_radius: float = 15.0
    min_food_distance: float = 80.0
    inhale_duration: int = 120
    exhale_duration: int = 150
    rest_duration: int = 60
    # not a reference kwarg: keep finished envs running instead of resetting them (salp_config_t.no_autoreset)
    no_autoreset: bool = False

    def __post_init__(self):
        # snake:36 base_num_food_items = max(0, num_food_items)
        self.num_food_items = max(0, int(self.num_food_items))
        if self.num_food_items > MAX_FOOD:
            raise ValueError(f"num_food_items={self.num_food_items} exceeds SALP_MAX_FOOD={MAX_FOOD}")
        if not 0 <= int(self.max_observed_food) <= MAX_OBSERVED_FOOD:
            raise ValueError(f"max_observed_food must be in [0, {MAX_OBSERVED_FOOD}]")
        if self.exhale_duration > 255 or self.inhale_duration > 255:
            raise ValueError("inhale/exhale durations above 255 steps do not fit the packed breathing word")

    @property
    def obs_dim(self) -> int:  # snake:79-80
        return 10 + 4 * int(self.max_observed_food) + 2

    @property
    def act_dim(self) -> int:  # snake:69-74
        return 1 if self.forced_breathing else 2

    def to_c(self) -> CConfig:
        c = CConfig()
        c.struct_size = ctypes.sizeof(CConfig)
        for name, _ in CConfig._fields_:
            if name in ("struct_size",):
                continue
            v = getattr(self, name)
            setattr(c, name, int(v) if isinstance(v, (bool, int)) else float(v))
        return c

    def env_kwargs(self) -> Dict[str, Any]:
        """The 13 reference kwargs (for handing the same parameters to the reference class)."""
        names = ("width", "height", "num_food_items", "food_reward", "collision_penalty", "time_penalty",
                 "efficiency_bonus", "forced_breathing", "max_observed_food", "random_food_count",
                 "respawn_food", "proximity_reward_weight", "max_steps_without_food")
        return {n: getattr(self, n) for n in names}


# environment.params of the reference's presets (configs/*.yaml), restated so the GPU box —
# which has no /root/reference — can name them.  `load_env_config(path)` reads a YAML file.
PRESETS: Dict[str, Dict[str, Any]] = {
    # configs/single_food.yaml:9-18
    "single_food": dict(num_food_items=1, food_reward=1000.0, collision_penalty=-10.0, time_penalty=-0.1,
                        proximity_reward_weight=5.0, respawn_food=True, forced_breathing=True,
                        max_steps_without_food=1500, efficiency_bonus=0.0),
    # configs/single_food_long_horizon.yaml:9-18
    "single_food_long_horizon": dict(num_food_items=1, food_reward=1000.0, collision_penalty=-500.0,
                                     time_penalty=-0.5, proximity_reward_weight=3.0, respawn_food=True,
                                     forced_breathing=True, max_steps_without_food=2000,
                                     efficiency_bonus=0.0),
    # configs/sac_gail.yaml:8-14 (identical to configs/defaults.yaml:9-15)
    "sac_gail": dict(num_food_items=12, food_reward=15.0, collision_penalty=-30.0, time_penalty=-0.05,
                     efficiency_bonus=2.0, forced_breathing=True),
}
PRESETS["defaults"] = dict(PRESETS["sac_gail"])


def load_env_config(name_or_path: str, **overrides) -> SalpSnakeConfig:
    """`name_or_path`: a preset name above, or a YAML file with an `environment:` block in the
    reference's schema (config_loader.py:83-115).  As in train.py:50-52 only
    `environment.params` reaches the env; width/height keep their 800x600 defaults."""
    if name_or_path in PRESETS:
        params = dict(PRESETS[name_or_path])
    else:
        if not os.path.isfile(name_or_path):
            raise FileNotFoundError(f"unknown preset or missing file: {name_or_path}")
        import yaml
        with open(name_or_path) as f:
            doc = yaml.safe_load(f)
        params = dict((doc.get("environment") or {}).get("params") or {})
    params.update(overrides)
    fields = {f.name for f in dataclasses.fields(SalpSnakeConfig)}
    unknown = set(params) - fields
    if unknown:
        raise TypeError(f"unexpected environment params: {sorted(unknown)}")
    return SalpSnakeConfig(**params)
