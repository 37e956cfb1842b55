import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/profiles')
import step_mode
for preset in ("sac_gail", "single_food_long_horizon"):
    for n in (4096, 65536, 262144):
        r = step_mode.run(n, preset=preset, iters=200); r["preset"] = preset
        print(json.dumps(r), flush=True)
