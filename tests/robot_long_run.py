#!/usr/bin/env python3
"""How far the HEAD-simulator kernel drifts from the C oracle over many cycles (tests/ pin <= 1e-6 over <= 12 cycles):
max relative state difference after every cycle, and which row / robot carries it."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import robot_oracle_lib as rol
from underwater_swimmer_rl_amd.robot_env import SalpRobotVectorEnv

n, seed, T = 2048, 5, 40
env = SalpRobotVectorEnv(n, device="cuda:0", seed=seed)
orc = rol.RobotOracleVec(n, seed=seed)
orc.reset(np.zeros(n, np.uint8))
rng = np.random.default_rng(2)
for t in range(T):
    a = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1, n), rng.uniform(-1, 1, n)], axis=1).astype(np.float32)
    obs, rew, term, trunc, info = env.step(a)
    ref = orc.step(a)
    assert np.array_equal(info["inner_steps"].cpu().numpy(), ref["inner_steps"])
    s, o = np.asarray(env.get_state()), orc.get_state()
    d = np.abs(s - o) / np.maximum(1.0, np.abs(o))
    row = int(np.nanargmax(np.nanmax(d, axis=1))); col = int(np.nanargmax(d[row]))
    print(f"cycle {t + 1:3d}: max rel diff {np.nanmax(d):.3e} (row {row}, robot {col}: {s[row, col]:.12g} vs {o[row, col]:.12g}); "
          f"per block pos {np.nanmax(d[0:3]):.1e} vel {np.nanmax(d[3:6]):.1e} euler {np.nanmax(d[6:9]):.1e} omega {np.nanmax(d[9:12]):.1e}", flush=True)
