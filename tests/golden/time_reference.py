#!/usr/bin/env python3
"""Times the REFERENCE's own SalpSnakeEnv.step (src/salp/environments/salp_snake_env.py:157-202 over the legacy
parent, scripts/utilities/salp_robot.py:119-156) on one host core, in the build container — the reference's Python
files do not travel to the GPU box, so this is where its CPU rate can be measured.

Workload = BASELINE.json configs[0] as SURVEY.md §8d defines it: 1 env, single_food.yaml parameters, 1000
random-action steps (actions U(-1, 1) from numpy.random.default_rng(123) as float32, `random.seed(0);
np.random.seed(0)`), reset on done.  The env runs on its own `random` / `numpy.random` draws here (the Philox
proxies of ref_harness.py are test plumbing and would be timed too).  Repeated for ~10 s; writes
profiles/reference_cpu_rate.json, which bench.py quotes in `cpu_baseline.python_reference`.
Run:  python tests/golden/time_reference.py
"""
import json
import os
import platform
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness as rh  # noqa: E402
import underwater_swimmer_rl_amd as pkg  # noqa: E402


def main():
    cls, _, _ = rh.load_reference()
    # undo the draw redirection: time the reference on its own generators
    sys.modules["salp.environments.salp_snake_env"].random = random
    sys.modules["salp.environments.salp_robot_env"].np = np
    params = pkg.load_env_config("single_food").env_kwargs()
    steps_per_run, budget_s = 1000, 10.0
    rates, runs, total_steps, resets = [], 0, 0, 0
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < budget_s:
        random.seed(0)
        np.random.seed(0)
        act = np.random.default_rng(123).uniform(-1, 1, size=(steps_per_run, 1)).astype(np.float32)
        env = cls(render_mode=None, **params)
        env.reset()
        t0 = time.perf_counter()
        for t in range(steps_per_run):
            obs, rew, term, trunc, info = env.step(act[t])
            if term or trunc:
                env.reset()
                resets += 1
        dt = time.perf_counter() - t0
        rates.append(steps_per_run / dt)
        runs += 1
        total_steps += steps_per_run
    rates.sort()
    out = {
        "what": "reference SalpSnakeEnv.step (pure Python), BASELINE configs[0]: 1 env, single_food.yaml, 1000 random-action steps, reset on done",
        "env_steps_per_s_median": rates[len(rates) // 2], "env_steps_per_s_min": rates[0], "env_steps_per_s_max": rates[-1],
        "runs": runs, "steps_per_run": steps_per_run, "resets_per_run": resets / runs, "cores": 1,
        "host": platform.node(), "machine": platform.machine(), "cpu": _cpu_model(), "python": platform.python_version(),
        "numpy": np.__version__, "measured_in": "build container (the reference cannot travel to the GPU box)",
        "script": "tests/golden/time_reference.py",
    }
    path = os.path.join(ROOT, "profiles", "reference_cpu_rate.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


if __name__ == "__main__":
    main()
