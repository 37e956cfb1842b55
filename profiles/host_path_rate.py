#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer surface (salp_vec_rollout with numpy buffers: H2D of the actions, the fused
kernel, D2H of every output, synchronous) — the number DESIGN.md §6 quotes beside the device-pointer bench value."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import underwater_swimmer_rl_amd as salp

n, H = 262144, 50
env = salp.SalpVectorEnv("single_food_long_horizon", num_envs=n, device="cuda:0", seed=0, output="numpy")
act = np.random.default_rng(0).uniform(-1, 1, (H, n, 1)).astype(np.float32)
env.rollout(act)
t0 = time.perf_counter(); reps = 4
for _ in range(reps):
    out = env.rollout(act)
dt = (time.perf_counter() - t0) / reps
b = 106.448 * n * H
print(json.dumps({"envs": n, "chunk": H, "s_per_launch": dt, "env_steps_per_s": n * H / dt, "GBps_over_pcie_and_host_copy": b / dt / 1e9,
                  "note": "pageable numpy buffers, synchronous hipMemcpy both ways"}))
