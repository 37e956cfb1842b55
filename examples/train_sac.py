#!/usr/bin/env python3
"""BASELINE.json configs[4]: SAC (configs/sac_gail.yaml agent block) with a PyTorch-ROCm policy on the HIP
VectorEnv; reports the wall-clock to the first food capture.  One GPU:
    python examples/train_sac.py --envs 4096 --steps 2000
Data-parallel over the GPUs of one node (one env shard, replay buffer and learner replica per rank, gradients
averaged over RCCL between the hipGraph segments of an iteration):
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_sac.py"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import underwater_swimmer_rl_amd as salp
from underwater_swimmer_rl_amd.sac import SAC, SACConfig, train_sac, train_sac_graphed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="sac_gail")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=2000, help="vector env steps")
    ap.add_argument("--learning-starts", type=int, default=50)
    ap.add_argument("--updates-per-step", type=int, default=1)
    ap.add_argument("--stop-at-first-food", action="store_true")
    ap.add_argument("--log-every", type=int, default=200)
    ap.add_argument("--eager", action="store_true", help="issue every kernel from Python (train_sac) instead of "
                    "one hipGraph replay per vector step (train_sac_graphed)")
    ap.add_argument("--graphed", action="store_true", help="under torchrun (more than one rank): use the segmented hipGraph form "
                    "(graph segments + RCCL gradient all-reduces between the replays).  The multi-rank default is --eager until that "
                    "form has run on two real GPUs (it is covered by gloo and world-size-1 tests only)")
    args = ap.parse_args()
    import torch
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    dev = f"cuda:{local}"
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device(dev))
    # --envs is per GPU; global env indices keep every env's draw stream distinct across ranks
    env = salp.SalpVectorEnv(args.preset, num_envs=args.envs, device=dev, seed=0, env_index_base=rank * args.envs)
    cfg = SACConfig.from_preset(args.preset)
    cfg.learning_starts, cfg.updates_per_step = args.learning_starts, args.updates_per_step
    agent = SAC(env.obs_dim, env.act_dim, cfg, device=dev, seed=0, data_parallel=world > 1,
                act_low=env.single_action_space.low, act_high=env.single_action_space.high)
    if world > 1 and not args.graphed:
        args.eager = True
    if args.eager:
        m = train_sac(env, agent, args.steps, log_every=args.log_every, stop_at_first_food=args.stop_at_first_food)
    else:
        m = train_sac_graphed(env, agent, args.steps, stop_at_first_food=args.stop_at_first_food)
    m["mode"] = "eager" if args.eager else "hipgraph"
    m["stats"] = env.stats()
    m["config"] = {"preset": args.preset, "envs_per_gpu": args.envs, "n_gpus": world, "batch_size": cfg.batch_size,
                   "gamma": cfg.gamma}
    if rank == 0:
        print(json.dumps(m))
    env.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
