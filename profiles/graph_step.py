"""Acting-loop rate (policy MLP 24-256-256-1 + salp_vec_step) at small batches: eager calls vs one
hipGraph of K steps (SalpVectorEnv.capture_policy_steps).  python3 profiles/graph_step.py"""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from underwater_swimmer_rl_amd import SalpVectorEnv

K = 64
for n in (256, 4096, 65536):
    env = SalpVectorEnv("single_food_long_horizon", num_envs=n, seed=1)
    dev = env.device
    net = torch.nn.Sequential(torch.nn.Linear(24, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(),
                              torch.nn.Linear(256, 1), torch.nn.Tanh()).to(dev)
    for p in net.parameters():
        p.requires_grad_(False)
    obs, _ = env.reset()
    for _ in range(20):
        obs, *_ = env.step(net(obs))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10 * K):
        obs, *_ = env.step(net(obs), want_final_observation=False)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / (10 * K)
    g = env.capture_policy_steps(net, n_steps=K, want_final_observation=False)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / (10 * K)
    print(json.dumps({"envs": n, "eager_us_per_step": round(eager * 1e6, 2), "graph_us_per_step": round(graph * 1e6, 2),
                      "eager_env_steps_per_s": round(n / eager), "graph_env_steps_per_s": round(n / graph)}), flush=True)
    env.close()
