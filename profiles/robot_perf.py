#!/usr/bin/env python3
"""Throughput of the batched HEAD simulator (salp_robot_step_kernel): env-steps/s (one env step = one
breathing cycle) and Euler steps/s, for a given cycle-length distribution."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from underwater_swimmer_rl_amd.robot_env import SalpRobotVectorEnv

def run(n, coast_hi, iters=6):
    env = SalpRobotVectorEnv(n, device="cuda:0", seed=0)
    g = torch.Generator(device="cuda").manual_seed(0)
    def actions():
        a = torch.rand((n, 3), generator=g, device="cuda")
        a[:, 1] *= coast_hi
        a[:, 2] = a[:, 2] * 2 - 1
        return a
    env.step(actions()); torch.cuda.synchronize()
    tot_ms, tot_inner, max_inner = 0.0, 0, 0
    for _ in range(iters):
        a = actions()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); obs, rew, term, trunc, info = env.step(a); e.record(); e.synchronize()
        tot_ms += s.elapsed_time(e)
        tot_inner += int(info["inner_steps"].sum()); max_inner = max(max_inner, int(info["inner_steps"].max()))
    env.close()
    sec = tot_ms / 1e3
    return {"envs": n, "coast_max_s": 10 * coast_hi, "ms_per_env_step_batch": tot_ms / iters, "env_steps_per_s": n * iters / sec,
            "euler_steps_per_s": tot_inner / sec, "mean_inner_steps": tot_inner / (n * iters), "max_inner_steps": max_inner}

if __name__ == "__main__":
    for sched in ("0", "1"):          # index order vs longest-cycle-first walk (robot_schedule_* in salp_robot.hip)
        os.environ["SALP_ROBOT_SCHEDULE"] = sched
        for n, c in ((65536, 0.1), (262144, 0.1), (262144, 1.0)):
            print(json.dumps({"schedule": int(sched), **run(n, c)}), flush=True)
