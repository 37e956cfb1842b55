#!/usr/bin/env python3
"""configs/sac_gail.yaml end to end (BASELINE.json configs[4]): SAC on the HIP VectorEnv with the GAIL
reward mix 0.3*r_env + 0.7*r_gail, discriminator trained against the reference's human demonstrations
(.npz extracts, tests/golden/human_demo_*.npz).  One GPU:
    python examples/train_sac_gail.py --envs 4096 --steps 1000
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import underwater_swimmer_rl_amd as salp
from underwater_swimmer_rl_amd.gail import Discriminator, ExpertBuffer, gail_reward_fn
from underwater_swimmer_rl_amd.sac import SAC, DeviceReplayBuffer, SACConfig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--demos", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--log-every", type=int, default=200)
    args = ap.parse_args()
    dev = "cuda:0"
    env = salp.SalpVectorEnv("sac_gail", num_envs=args.envs, device=dev, seed=0)
    cfg = SACConfig.from_preset("sac_gail")            # lr 3e-4, batch 128, buffer 5e5, gamma .99, tau .005, alpha .1
    cfg.learning_starts = 50                            # training.start_training_after = 500 single-env steps
    agent = SAC(env.obs_dim, env.act_dim, cfg, device=dev, seed=0)
    disc = Discriminator(env.obs_dim, env.act_dim, (256, 256), learning_rate=3e-4, device=dev)   # gail.discriminator_lr
    experts = ExpertBuffer(env.obs_dim, env.act_dim, device=dev)
    n_demo = experts.load_directory(args.demos, "human_demo_*.npz")
    assert n_demo >= 5, "gail.min_expert_episodes = 5"
    buf = DeviceReplayBuffer(cfg.buffer_size, env.obs_dim, env.act_dim, dev)
    mix = gail_reward_fn(disc, 0.3, 0.7)                # gail.reward_env_weight / reward_gail_weight
    low = torch.as_tensor(env.single_action_space.low, device=dev)
    high = torch.as_tensor(env.single_action_space.high, device=dev)
    obs, _ = env.reset()
    obs = obs.clone()
    t0 = time.perf_counter()
    first_food = None
    last, dlast = {}, {}
    for step in range(args.steps):
        act = agent.act(obs) if step >= cfg.learning_starts else low + (high - low) * torch.rand((args.envs, env.act_dim), device=dev)
        nobs, rew, term, trunc, info = env.step(act)
        done = term | trunc
        next_obs = torch.where(done[:, None], info["final_observation"], nobs)
        buf.add(obs, act, mix(obs, act, rew), next_obs, term)
        if first_food is None and bool((info["food_collected"] > 0).any()):
            torch.cuda.synchronize()
            first_food = (time.perf_counter() - t0, step + 1)
        if step >= cfg.learning_starts:
            o, a, _, _, _ = buf.sample(cfg.batch_size)
            dlast = disc.update(experts.sample(cfg.batch_size), {"observations": o, "actions": a})   # update_freq 1
            last = agent.update(buf.sample(cfg.batch_size))
        obs = nobs.clone()
        if args.log_every and (step + 1) % args.log_every == 0:
            print(f"[sac+gail] step {step + 1} D acc {float(dlast.get('discriminator_accuracy', float('nan'))):.3f} "
                  f"critic {float(last.get('critic_loss', float('nan'))):.3f}", flush=True)
    torch.cuda.synchronize()
    out = {"wall_s": time.perf_counter() - t0, "vector_steps": args.steps, "env_steps": args.steps * args.envs,
           "first_food_wall_s": first_food[0] if first_food else None,
           "first_food_vector_step": first_food[1] if first_food else None, "expert_pairs": len(experts),
           "discriminator": Discriminator.metrics_to_host(dlast) if dlast else {}, "sac": {k: float(v) for k, v in last.items()}, "stats": env.stats()}
    print(json.dumps(out))
    env.close()


if __name__ == "__main__":
    main()
