"""Navigation evaluation harness: metric maths on CPU, trials on the GPU."""
import numpy as np
import pytest

from underwater_swimmer_rl_amd.navigation_eval import navigation_config, navigation_metrics, summarize


def test_metrics_on_known_paths():
    start, goal = np.array([150.0, 300.0]), np.array([650.0, 300.0])
    T = 100
    pos = np.zeros((T + 1, 2, 2))
    # trial 0: straight line that stops 40 px short of the goal after 92 steps (then frozen)
    xs = np.minimum(150.0 + 5.0 * np.arange(T + 1), 610.0)
    pos[:, 0, 0], pos[:, 0, 1] = xs, 300.0
    # trial 1: same x, offset 30 px sideways for the whole run, never gets within 50 px
    pos[:, 1, 0], pos[:, 1, 1] = np.minimum(150.0 + 4.0 * np.arange(T + 1), 550.0), 330.0
    steps = np.array([92, 100])
    m = navigation_metrics(pos, steps, start, goal)
    assert m["success"].tolist() == [True, False]
    assert abs(m["path_length"][0] - 500.0) < 1e-9 and abs(m["path_ratio"][0] - 1.0) < 1e-12
    assert abs(m["straightness"][0] - 1.0) < 1e-12 and m["lateral_deviation"][0] == 0.0
    assert abs(m["lateral_deviation"][1] - 30.0) < 1e-12
    assert abs(m["final_distance"][1] - np.hypot(100.0, 30.0)) < 1e-9
    s = summarize(m)
    assert s["num_trials"] == 2 and s["success_rate"] == 0.5 and s["avg_steps"] == 96.0


def test_navigation_config_is_the_reference_eval_env():
    c = navigation_config()   # eval/collect_navigation_data.py:62-70
    assert (c.num_food_items, c.forced_breathing, c.respawn_food, c.max_steps_without_food) == (1, True, False, 3000)
    assert c.food_reward == 10.0 and c.collision_penalty == -50.0


@pytest.mark.gpu
def test_pursuit_policy_reaches_the_goal_and_matches_the_oracle():
    import torch
    import oracle_lib as ol
    from underwater_swimmer_rl_amd import _capi
    from underwater_swimmer_rl_amd.navigation_eval import pursuit_policy, run_navigation_trials
    m = run_navigation_trials(pursuit_policy(), num_trials=256, max_steps=3000, seed=3, heading_seed=1)
    s = summarize(m)
    assert s["success_rate"] > 0.5, s
    assert np.isfinite(m["path_ratio"]).all() and (m["path_ratio"][m["success"]] >= 0.89).all()
    assert (m["steps"] <= 3000).all() and (m["steps"][m["success"]] < 3000).all()
    print("pursuit baseline:", {k: round(v, 3) if isinstance(v, float) else v for k, v in s.items()})
