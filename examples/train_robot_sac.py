#!/usr/bin/env python3
"""The HEAD simulator's training script (src/salp/environments/train_robot.py: SAC "MlpPolicy", lr 3e-4, buffer 1e5,
batch 512, ent_coef "auto", gamma 0.99, tau 0.005, 8 SubprocVecEnv workers) on the batched HIP simulator:
    python examples/train_robot_sac.py --envs 4096 --steps 300
One env step is one whole breathing cycle of every robot (contract, jet, coast)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_swimmer_rl_amd.robot_env import SalpRobotVectorEnv
from underwater_swimmer_rl_amd.sac import SAC, SACConfig, train_sac, train_sac_graphed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=300, help="vector env steps (breathing cycles per robot)")
    ap.add_argument("--learning-starts", type=int, default=10)
    ap.add_argument("--eager", action="store_true")
    args = ap.parse_args()
    env = SalpRobotVectorEnv(args.envs, device="cuda:0", seed=0)
    cfg = SACConfig.from_preset("salp_robot")
    cfg.learning_starts = args.learning_starts
    agent = SAC(env.obs_dim, env.act_dim, cfg, device="cuda:0", seed=0,
                act_low=env.single_action_space.low, act_high=env.single_action_space.high)
    run = train_sac if args.eager else train_sac_graphed
    m = run(env, agent, args.steps)
    m["first_target_reached_wall_s"] = m.pop("first_food_wall_s")
    m["first_target_reached_vector_step"] = m.pop("first_food_vector_step")
    m["config"] = {"envs": args.envs, "batch_size": cfg.batch_size, "buffer_size": cfg.buffer_size}
    print(json.dumps(m))
    env.close()


if __name__ == "__main__":
    main()
