"""Soak run (not collected by pytest): crowded tanks — many foods, a large minimum distance, small tanks — where resets and
respawns end in the reference's fallback placement (snake:120-131, :270-276), foods land next to the swimmer and several sit
inside the capture radius at once; shaping weight 5.   python3 tests/soak_crowded.py [cases]"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import test_gpu_parity as T
import oracle_lib as ol
import underwater_swimmer_rl_amd as pkg
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for case in range(cases):
    rng = np.random.default_rng(8000 + case)
    F = int(rng.choice([4, 8, 12, 16]))
    std_tank = bool(rng.random() < 0.4)
    kw = dict(num_food_items=F, proximity_reward_weight=5.0, min_food_distance=float(rng.uniform(90, 160)),
              max_steps_without_food=int(rng.integers(100, 500)), respawn_food=bool(rng.random() < 0.8),
              random_food_count=bool(rng.random() < 0.3), forced_breathing=bool(rng.random() < 0.7))
    if not std_tank:
        kw.update(width=int(rng.integers(420, 700)), height=int(rng.integers(400, 600)))
    cfg = pkg.load_env_config("sac_gail", **kw)
    n, H, seed = 4096 + int(rng.integers(0, 100)), 1100, int(rng.integers(0, 2 ** 31))
    act = T.make_actions(cfg, H, n, seed=case, scale=1.1)
    try:
        got, dev = T.run_device(cfg, n, act, seed=seed, want_final=bool(case & 1))
        orc = ol.OracleVec(cfg, n, seed=seed, threads=16)
        ref = orc.rollout(act, want_final=bool(case & 1))
        d = T.assert_parity(cfg, got, ref, f"case {case}")
        T.assert_state_parity(cfg, dev, orc, f"case {case}")
        st = dev.stats()
        print(f"case {case}: F={F} {cfg.width}x{cfg.height} min_dist {kw['min_food_distance']:.0f} max obs diff {d[0]:.2e} reward {d[1]:.2e} "
              f"episodes {st['episodes']} food {st['food_collected']}", flush=True)
        dev.close(); orc.close()
    except AssertionError as e:
        bad += 1
        print("FAIL", case, kw, str(e)[:300], flush=True)
print("done, failures:", bad)
