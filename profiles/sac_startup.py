import time, sys, os
sys.path.insert(0, os.getcwd())
t00 = time.perf_counter()
import torch
t_imp = time.perf_counter()
import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd.sac import SAC, SACConfig, train_sac_graphed, DeviceReplayBuffer
def T(label, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); print(f"{label:40s} {1e3*(time.perf_counter()-t):9.1f} ms", flush=True); return r
print("import torch", 1e3*(t_imp-t00))
T("cuda init (first tensor)", lambda: torch.zeros(1, device="cuda"))
env = T("env create", lambda: pkg.SalpVectorEnv("sac_gail", num_envs=4096, device="cuda:0", seed=0))
cfg = SACConfig.from_preset("sac_gail"); cfg.learning_starts = 50
agent = T("agent create", lambda: SAC(env.obs_dim, env.act_dim, cfg, device="cuda:0", seed=0, act_low=env.single_action_space.low, act_high=env.single_action_space.high))
obs, _ = T("env.reset", lambda: env.reset())
act = T("torch.rand act", lambda: torch.rand((4096, env.act_dim), device="cuda") * 2 - 1)
out = T("first env.step", lambda: env.step(act))
out = T("second env.step", lambda: env.step(act))
a = T("first agent.act", lambda: agent.act(obs))
a = T("second agent.act", lambda: agent.act(obs))
buf = T("buffer create", lambda: DeviceReplayBuffer(cfg.buffer_size, env.obs_dim, env.act_dim, "cuda:0"))
def add():
    nobs, rew, term, trunc, info = out
    buf.add_capturable(obs, act, rew, nobs, term); buf.advance_host(4096)
T("first buffer add", add)
T("second buffer add", add)
b = T("first sample", lambda: buf.sample_capturable(cfg.batch_size))
T("first stage_critic", lambda: agent.stage_critic(b))
T("first stage_actor", lambda: agent.stage_actor())
T("first stage_finish", lambda: agent.stage_finish())
b = buf.sample_capturable(cfg.batch_size)
T("second stage_critic", lambda: agent.stage_critic(b))
T("second stage_actor", lambda: agent.stage_actor())
T("second stage_finish", lambda: agent.stage_finish())
env.close()
env = pkg.SalpVectorEnv("sac_gail", num_envs=4096, device="cuda:0", seed=0)
agent = SAC(env.obs_dim, env.act_dim, cfg, device="cuda:0", seed=0, act_low=env.single_action_space.low, act_high=env.single_action_space.high)
m = T("train_sac_graphed to first food (warm process)", lambda: train_sac_graphed(env, agent, 600, stop_at_first_food=True))
print({k: m[k] for k in ("first_food_wall_s", "first_food_vector_step", "learn_ms_per_vector_step")})
