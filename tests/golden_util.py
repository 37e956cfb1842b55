"""Shared helpers for the golden-vector tests (tests/golden/ref_*.npz, made by gen_golden.py)."""
import glob
import json
import os

import numpy as np

import underwater_swimmer_rl_amd as pkg

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KW = ("width", "height", "num_food_items", "food_reward", "collision_penalty", "time_penalty", "efficiency_bonus",
      "forced_breathing", "max_observed_food", "random_food_count", "respawn_food", "proximity_reward_weight",
      "max_steps_without_food")


def fixture_names():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "ref_*.npz")))


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"ref_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["cfg_json"]))
    cfg = pkg.SalpSnakeConfig(**{k: meta[k] for k in KW}, no_autoreset=bool(meta.get("no_autoreset", False)))
    return z, meta, cfg


def angle_cols(cfg):
    return [4] + [10 + 4 * s + 3 for s in range(cfg.max_observed_food)]


def obs_diff(cfg, a, b):
    """|a - b| with the angle/pi columns compared on the circle (-1 == +1)."""
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    for c in angle_cols(cfg):
        d[..., c] = np.minimum(d[..., c], 2.0 - d[..., c])
    return np.nan_to_num(d, nan=0.0)
