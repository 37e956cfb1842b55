set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_v13.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v13.log; exit 1; }
tail -2 gpurun_out/r02/gpu_tests_v13.log
python - <<'PY'
import json, torch, time
import underwater_swimmer_rl_amd as salp
for F in (5, 8, 12):
    env = salp.SalpVectorEnv("sac_gail", num_envs=262144, device="cuda:0", seed=0, num_food_items=F)
    act = torch.rand((250, 262144, 1), device="cuda") * 2 - 1
    for _ in range(6): env.rollout(act)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): env.rollout(act)
    e.record(); torch.cuda.synchronize()
    print(json.dumps({"foods": F, "ms_per_launch": s.elapsed_time(e) / 10}))
    env.close()
PY
