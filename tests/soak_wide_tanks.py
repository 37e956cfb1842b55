"""Soak run (not collected by pytest): wide tanks (1100..2000 px: fp32 positions carry up to 1.2e-4 px of rounding) with the
largest shaping weight and no other reward term — the worst case for the fp32 bearing of the multi-food kernels' reward.
Prints the largest reward and observation differences per case.   python3 tests/soak_wide_tanks.py [cases]"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import test_gpu_parity as T
import oracle_lib as ol
import underwater_swimmer_rl_amd as pkg
from golden_util import obs_diff
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worst_r = worst_o = 0.0
for case in range(cases):
    rng = np.random.default_rng(7000 + case)
    F = int(rng.choice([3, 5, 8, 12, 16]))
    cfg = pkg.load_env_config("sac_gail", num_food_items=F, width=int(rng.integers(1100, 2000)), height=int(rng.integers(900, 1500)),
                              proximity_reward_weight=5.0, time_penalty=0.0, efficiency_bonus=0.0,
                              max_steps_without_food=int(rng.integers(200, 600)))
    n, H, seed = 4096, 700, int(rng.integers(0, 2 ** 31))
    act = T.make_actions(cfg, H, n, seed=case)
    got, dev = T.run_device(cfg, n, act, seed=seed)
    orc = ol.OracleVec(cfg, n, seed=seed, threads=16)
    ref = orc.rollout(act)
    assert np.array_equal(got["terminated"], ref["terminated"]) and np.array_equal(got["truncated"], ref["truncated"])
    r = ref["reward"]
    dr = float((np.abs(got["reward"] - r) / np.maximum(1.0, np.abs(r))).max())
    do = float(obs_diff(cfg, got["obs"], ref["obs"]).max())
    worst_r, worst_o = max(worst_r, dr), max(worst_o, do)
    print(f"case {case}: F={F} {cfg.width}x{cfg.height} max rel reward diff {dr:.3e} max obs diff {do:.3e}", flush=True)
    dev.close(); orc.close()
print(f"done: worst reward {worst_r:.3e} worst obs {worst_o:.3e}")
