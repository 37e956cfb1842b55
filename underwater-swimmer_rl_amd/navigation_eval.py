"""Fixed start -> goal navigation trials on the batched simulator.

Restates the protocol and metrics of the reference's eval/collect_navigation_data.py:
  * env: one food, forced breathing, respawn off, max_steps_without_food = 3000 (:62-70);
  * per trial: reset, then overwrite pose (start position, zero velocity, heading ~ U(-pi, pi), zero
    angular velocity), place the single food at the goal, zero steps_since_food (:76-89) — here through
    `set_state`;
  * roll the policy until the swimmer is within `goal_radius` (50 px) of the goal or `max_steps` (:97-114); the
    reference's loop ignores `terminated` / `truncated`, so wall contacts do not end a trial (`no_autoreset`);
  * metrics (:117-196): path length (+ final distance to the goal), success (final distance < 50),
    path ratio, straightness, mean lateral deviation from the start-goal line, bounding-box area and
    area ratio, x / y range, and the spline-smoothed path ratio (:138-165, `spline_path_length`).
All trials run at once: trial i is env i.  The only published numbers for this protocol are in
eval/results/navigation_stats_20251207_165158.json (a trained SB3 policy: success 1.00,
1773.78 +/- 253.8 steps, path ratio 1.173, straightness 0.863).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np

from . import _capi
from .config import SalpSnakeConfig
from .vector_env import SalpVectorEnv


def navigation_config(**overrides):
    # no_autoreset: the reference's trial loop ignores `done` (:97-114) — a swimmer that touches a wall is clamped,
    # bounces and swims on; nothing is reset until the trial ends
    params = dict(num_food_items=1, forced_breathing=True, respawn_food=False, max_steps_without_food=3000,
                  no_autoreset=True)
    params.update(overrides)
    return SalpSnakeConfig(**params)   # the reference builds SalpSnakeEnv with its class defaults here


def pursuit_policy(gain: float = 3.0) -> Callable:
    """A scripted baseline: steer the nozzle against the relative bearing of the food (obs column 13).
    A positive nozzle angle yields a negative torque (legacy:285), hence the minus sign."""
    def policy(obs):
        return (-gain * obs[:, 13:14]).clamp(-1.0, 1.0)
    return policy


def run_navigation_trials(policy: Callable, num_trials: int = 100, start_pos=(150.0, 300.0), goal_pos=(650.0, 300.0),
                          max_steps: int = 3000, goal_radius: float = 50.0, device="cuda:0", seed: int = 0,
                          heading_seed: int = 0, env: Optional[SalpVectorEnv] = None) -> Dict[str, np.ndarray]:
    import torch
    cfg = navigation_config()
    own = env is None
    env = env or SalpVectorEnv(cfg, num_trials, device=device, seed=seed)
    n = env.num_envs
    env.reset()
    f64, i32 = env.get_state()
    rng = np.random.default_rng(heading_seed)
    f64[_capi.F_X], f64[_capi.F_Y] = start_pos
    f64[_capi.F_VX] = 0.0
    f64[_capi.F_VY] = 0.0
    f64[_capi.F_THETA] = rng.uniform(-np.pi, np.pi, n)       # np.random.uniform(-pi, pi), :82
    f64[_capi.F_OMEGA] = 0.0
    f64[_capi.F_FOOD0], f64[_capi.F_FOOD0 + 1] = goal_pos
    i32[_capi.I_STEPS_SINCE_FOOD] = 0
    env.set_state(f64, i32)
    obs = env.observe().clone()

    dev = obs.device
    W, H = float(cfg.width), float(cfg.height)
    goal = torch.tensor(goal_pos, device=dev, dtype=torch.float32)
    pos = torch.empty((max_steps + 1, n, 2), device=dev)
    pos[0] = torch.tensor(start_pos, device=dev, dtype=torch.float32)
    running = torch.ones(n, dtype=torch.bool, device=dev)
    steps = torch.zeros(n, dtype=torch.int32, device=dev)
    collided = torch.zeros(n, dtype=torch.bool, device=dev)
    T = 0
    for t in range(max_steps):
        act = policy(obs)
        nobs, rew, term, trunc, info = env.step(act)
        # position of the step just taken (x / W, y / H in columns 0, 1); `terminated` / `truncated` are ignored as in
        # the reference's loop, the env runs on (no_autoreset)
        p = torch.stack([nobs[:, 0] * W, nobs[:, 1] * H], dim=1)
        pos[t + 1] = torch.where(running[:, None], p, pos[t])
        steps += running.to(torch.int32)
        collided |= running & (info["collision"] > 0)
        reached = (pos[t + 1] - goal).norm(dim=1) < goal_radius
        running = running & ~reached
        obs = nobs.clone()
        T = t + 1
        if not bool(running.any()):
            break
    pos_h = pos[: T + 1].cpu().numpy().astype(np.float64)
    steps_h = steps.cpu().numpy()
    out = navigation_metrics(pos_h, steps_h, np.asarray(start_pos, float), np.asarray(goal_pos, float), goal_radius)
    out["collided"] = collided.cpu().numpy()
    if own:
        env.close()
    return out


def spline_path_length(positions: np.ndarray, goal: np.ndarray) -> float:
    """The smoothed path length of eval/collect_navigation_data.py:138-165 for ONE trial (`positions` [L, 2], the
    recorded positions start..stop): consecutive duplicates removed, ~20 control points (every len // 20-th point,
    the last one always kept), a cubic smoothing spline through them (`splprep(s=500, k=3)`), its length over 500
    samples plus the distance from its last sample to the goal.  NaN when fewer than 4 points are left (the
    reference reports None) or scipy is missing."""
    try:
        from scipy.interpolate import splev, splprep
    except ImportError:
        return float("nan")
    keep = np.ones(len(positions), bool)
    keep[1:] = np.any(positions[1:] != positions[:-1], axis=1)
    uniq = positions[keep]
    if len(uniq) < 4:
        return float("nan")
    ctrl = uniq[::max(1, len(uniq) // 20)]
    if not np.array_equal(ctrl[-1], uniq[-1]):
        ctrl = np.vstack([ctrl, uniq[-1]])
    if len(ctrl) < 4:
        return float("nan")
    try:
        tck, _ = splprep([ctrl[:, 0], ctrl[:, 1]], s=500.0, k=3)
        sp = np.column_stack(splev(np.linspace(0, 1, 500), tck))
    except Exception:   # noqa: BLE001 — the reference swallows fit failures too (:164)
        return float("nan")
    return float(np.linalg.norm(np.diff(sp, axis=0), axis=1).sum() + np.linalg.norm(sp[-1] - goal))


def navigation_metrics(pos: np.ndarray, steps: np.ndarray, start: np.ndarray, goal: np.ndarray,
                       goal_radius: float = 50.0) -> Dict[str, np.ndarray]:
    """pos: [T+1, N, 2] with the position frozen after a trial stopped; steps: [N] steps taken."""
    n = pos.shape[1]
    optimal = float(np.linalg.norm(goal - start))
    seg = np.linalg.norm(np.diff(pos, axis=0), axis=2)            # frozen tail contributes zero length
    final = np.stack([pos[steps[i], i] for i in range(n)])
    final_distance = np.linalg.norm(final - goal, axis=1)
    path_length = seg.sum(axis=0) + final_distance                 # :130-132
    d = goal - start
    dn = d / (np.linalg.norm(d) + 1e-12)
    rel = pos - start
    lateral = np.abs(rel[..., 0] * dn[1] - rel[..., 1] * dn[0])    # perpendicular distance to the line
    lat_mean = np.array([lateral[: steps[i] + 1, i].mean() for i in range(n)])
    xmin, xmax = pos[..., 0].min(axis=0), pos[..., 0].max(axis=0)
    ymin, ymax = pos[..., 1].min(axis=0), pos[..., 1].max(axis=0)
    area = (xmax - xmin) * (ymax - ymin)
    spline_len = np.array([spline_path_length(pos[: steps[i] + 1, i], goal) for i in range(n)])
    return {
        "steps": steps.astype(np.int64), "path_length": path_length, "final_distance": final_distance,
        "success": final_distance < goal_radius, "path_ratio": path_length / optimal,
        "straightness": optimal / np.maximum(path_length, 1e-12), "lateral_deviation": lat_mean,
        "area_covered": area, "area_ratio": area / (optimal * goal_radius * 2), "x_range": xmax - xmin,
        "y_range": ymax - ymin, "optimal_distance": np.full(n, optimal),
        "spline_path_length": spline_len, "spline_path_ratio": spline_len / optimal,
    }


def summarize(m: Dict[str, np.ndarray]) -> Dict[str, float]:
    """The aggregate fields of eval/results/navigation_stats_*.json."""
    ok = m["success"]
    return {
        "num_trials": int(len(ok)), "success_rate": float(ok.mean()), "successful_trials": int(ok.sum()),
        "avg_path_length": float(m["path_length"].mean()), "std_path_length": float(m["path_length"].std()),
        "avg_path_ratio": float(m["path_ratio"].mean()),
        "avg_spline_path_ratio": float(np.nanmean(m["spline_path_ratio"])) if np.isfinite(m["spline_path_ratio"]).any() else None,
        "avg_straightness": float(m["straightness"].mean()),
        "std_straightness": float(m["straightness"].std()), "avg_steps": float(m["steps"].mean()),
        "std_steps": float(m["steps"].std()), "avg_lateral_deviation": float(m["lateral_deviation"].mean()),
        "avg_final_distance": float(m["final_distance"].mean()), "avg_area_covered": float(m["area_covered"].mean()),
        "avg_area_ratio": float(m["area_ratio"].mean()), "avg_x_range": float(m["x_range"].mean()),
        "avg_y_range": float(m["y_range"].mean()),
    }
