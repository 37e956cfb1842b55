#!/usr/bin/env python3
"""Generates tests/golden/nav_metrics.npz: the trial metrics of the reference's eval/collect_navigation_data.py
(`NavigationDataCollector.run_single_trial`, :73-196 — path length, path ratio, straightness, lateral deviation,
area, and the spline-smoothed path ratio of :138-165) on scripted paths, computed by the REFERENCE's own method.

The reference module is loaded by path; `stable_baselines3` (absent here) and the `src.salp...` import of its header
are satisfied by empty in-memory stand-ins (no arithmetic lives in them), the collector object is created without
its constructor (which loads a model file), and `run_single_trial` is driven by a stand-in env that replays a
scripted list of positions and a stand-in model that returns a zero action — so every number in the fixture comes
out of the reference's metric code.  Run:  python tests/golden/gen_nav_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SALP_REFERENCE", "/root/reference")


def load_collector():
    sb3 = types.ModuleType("stable_baselines3")
    sb3.SAC = type("SAC", (), {})
    sys.modules.setdefault("stable_baselines3", sb3)
    for name in ("src", "src.salp", "src.salp.environments"):
        sys.modules.setdefault(name, types.ModuleType(name))
    m = types.ModuleType("src.salp.environments.salp_snake_env")
    m.SalpSnakeEnv = type("SalpSnakeEnv", (), {})
    sys.modules.setdefault("src.salp.environments.salp_snake_env", m)
    spec = importlib.util.spec_from_file_location("ref_collect_navigation_data", os.path.join(REF, "eval", "collect_navigation_data.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.NavigationDataCollector


class ReplayEnv:
    """Replays a scripted path: robot_pos after step t is path[t + 1]."""

    def __init__(self, path):
        self.path, self.t = np.asarray(path, float), 0
        self.robot_pos = self.path[0].copy()
        self.robot_velocity = np.zeros(2)
        self.robot_angle = 0.0
        self.robot_angular_velocity = 0.0
        self.food_positions = []
        self.steps_since_food = 0

    def reset(self):
        self.t = 0
        return np.zeros(24, np.float32), {}

    def _get_extended_observation(self):
        return np.zeros(24, np.float32)

    def step(self, action):
        self.t += 1
        self.robot_pos = self.path[min(self.t, len(self.path) - 1)].copy()
        return np.zeros(24, np.float32), 0.0, False, False, {}


class ZeroModel:
    def predict(self, obs, deterministic=True):
        return np.zeros(1, np.float32), None


def scripted_paths(start, goal):
    rng = np.random.default_rng(5)
    paths, max_steps = [], []
    t = np.linspace(0, 1, 400)[:, None]
    line = start + (goal - start) * t
    paths.append(line + np.c_[np.zeros(400), 60 * np.sin(3 * np.pi * t[:, 0])]); max_steps.append(3000)      # reaches the goal radius
    stall = np.repeat(line[::4] + rng.normal(0, 1.5, (100, 2)), 4, axis=0)                                   # consecutive duplicates
    paths.append(stall); max_steps.append(3000)
    paths.append(np.array([start, start + [5.0, 1.0], start + [9.0, -2.0]])); max_steps.append(2)           # < 4 points: no spline
    far = start + (goal - start) * np.linspace(0, 0.55, 300)[:, None] + np.c_[np.zeros(300), 25 * np.cos(np.linspace(0, 9, 300))]
    paths.append(far); max_steps.append(299)                                                                 # never reaches the goal
    ang = np.linspace(0, 4 * np.pi, 500)
    loop = start + np.c_[np.linspace(0, 1, 500) * (goal - start)[0] + 40 * np.sin(ang), 40 * (1 - np.cos(ang))]
    paths.append(loop); max_steps.append(3000)                                                               # two loops on the way
    paths.append(start + rng.normal(0, 0.2, (40, 2)).cumsum(axis=0)); max_steps.append(39)                  # short random walk
    return paths, max_steps


def main():
    Collector = load_collector()
    start, goal = np.array([150.0, 300.0]), np.array([650.0, 300.0])
    c = object.__new__(Collector)
    c.start_pos, c.goal_pos, c.goal_radius = start.copy(), goal.copy(), 50
    c.optimal_distance = np.linalg.norm(goal - start)
    c.model = ZeroModel()
    keys = ["steps", "path_length", "path_ratio", "spline_path_length", "spline_path_ratio", "straightness", "final_distance",
            "success", "lateral_deviation", "area_covered", "area_ratio", "x_range", "y_range"]
    paths, max_steps = scripted_paths(start, goal)
    out = {k: [] for k in keys}
    traj = []
    for p, ms in zip(paths, max_steps):
        p = np.asarray(p, float)
        p[0] = start                      # run_single_trial starts its record at start_pos (:91)
        r = c.run_single_trial(ReplayEnv(p), max_steps=ms)
        for k in keys:
            v = r[k]
            out[k].append(np.nan if v is None else float(v))
        traj.append(np.asarray(r["positions"], float))
    T = max(len(t) for t in traj)
    pos = np.stack([np.vstack([t, np.repeat(t[-1:], T - len(t), axis=0)]) for t in traj], axis=1)   # [T, P, 2], frozen tails
    np.savez(os.path.join(HERE, "nav_metrics.npz"), pos=pos, start=start, goal=goal, goal_radius=np.float64(50.0),
             **{k: np.array(v) for k, v in out.items()})
    for k in ("steps", "path_ratio", "spline_path_ratio", "lateral_deviation"):
        print(k, np.round(out[k], 4))


if __name__ == "__main__":
    main()
