import json, torch, sys, os
sys.path.insert(0, os.getcwd())
import underwater_swimmer_rl_amd as salp
for name, kw in (("std_F1_K3", {}), ("generic_F1_K2", dict(max_observed_food=2)), ("generic_F1_width801", dict(width=801)), ("generic_F12_K2", dict(num_food_items=12, max_observed_food=2))):
    env = salp.SalpVectorEnv("single_food_long_horizon", num_envs=262144, device="cuda:0", seed=0, **kw)
    act = torch.rand((250, 262144, 1), device="cuda") * 2 - 1
    for _ in range(4): env.rollout(act)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(6): env.rollout(act)
    e.record(); torch.cuda.synchronize()
    print(json.dumps({"config": name, "ms_per_launch": s.elapsed_time(e) / 6, "obs_dim": env.obs_dim}), flush=True)
    env.close()
