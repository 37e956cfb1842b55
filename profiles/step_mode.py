#!/usr/bin/env python3
"""Step-per-launch (H = 1) timing of salp_vec_step on device pointers: kernel time by HIP events
over a batch of launches, and host wall per call (ctypes + launch)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import underwater_swimmer_rl_amd as salp

def run(n, preset="single_food_long_horizon", iters=300):
    env = salp.SalpVectorEnv(preset, num_envs=n, device="cuda:0", seed=0)
    act = torch.rand((n, env.act_dim), device="cuda") * 2 - 1
    for _ in range(20): env.step(act, want_final_observation=False)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(iters): env.step(act, want_final_observation=False)
    e.record(); host_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    dev_ms = s.elapsed_time(e) / iters
    cfg = env.cfg
    S = 48 + 8 * cfg.num_food_items
    bpe = 2 * S + 4 * cfg.act_dim + 4 * cfg.obs_dim + 4 + 2
    out = {"envs": n, "us_per_step_device": dev_ms * 1e3, "us_per_call_host_issue": host_issue / iters * 1e6,
           "env_steps_per_s": n / (wall / iters), "algorithmic_GBps": bpe * n / (dev_ms * 1e-3) / 1e9, "bytes_per_env_step": bpe}
    env.close()
    return out

if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 65536, 262144, 1048576]
    for n in sizes:
        print(json.dumps(run(n)), flush=True)
