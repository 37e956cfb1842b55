#!/usr/bin/env python3
"""Debug helper: one weird action at a time on the HEAD-simulator kernel, each in its own subprocess with a timeout."""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"big": "(1.0e9, 1.0e9, 0.0)", "coast_inf": "(0.5, float('inf'), 0.0)", "contr_inf": "(float('inf'), 0.0, 0.0)",
         "nan": "(float('nan'),) * 3", "coast_ninf": "(0.5, -float('inf'), 0.0)", "contr_2": "(2.0, 0.1, 0.0)", "contr_100": "(100.0, 0.1, 0.0)"}
if len(sys.argv) > 1:
    import numpy as np, torch, time
    sys.path.insert(0, ROOT)
    from underwater_swimmer_rl_amd.robot_env import SalpRobotVectorEnv
    env = SalpRobotVectorEnv(64, device="cuda:0", seed=3)
    a = np.tile(np.array([0.5, 0.1, 0.0], np.float32), (64, 1))
    a[0] = eval(CASES[sys.argv[1]])
    t0 = time.time()
    obs, rew, term, trunc, info = env.step(a)
    torch.cuda.synchronize()
    print(sys.argv[1], "ok", round(time.time() - t0, 3), "s, inner steps", info["inner_steps"].cpu().numpy()[:2], flush=True)
else:
    for k in CASES:
        try:
            r = subprocess.run([sys.executable, __file__, k], timeout=40, capture_output=True, text=True)
            print(k, "rc", r.returncode, r.stdout.strip()[-200:], r.stderr.strip()[-300:], flush=True)
        except subprocess.TimeoutExpired:
            print(k, "TIMEOUT (hang)", flush=True)
            break
