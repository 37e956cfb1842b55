"""Soak run (not collected by pytest) of the UNPREDICATED main-launch kernels — the register-food (2..12 slots),
16-slot (13..16 foods) and one-food STD instantiations, all three output signatures — against the oracle, at batch
sizes above the small-batch threshold (n x H > 2^22, so the range splits into the main launch over whole wavefronts and
the ragged tail).  Random reference-constant configurations: every flag, reward and time-out setting of
test_gpu_parity._random_cfg with K = 3; from case 30 on also other tanks / physics
(the STD = false kernels).   python3 tests/soak_main_kernels.py [cases]   (last run: 0 failures)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import test_gpu_parity as T   # noqa: E402
import oracle_lib as ol       # noqa: E402
import underwater_swimmer_rl_amd as pkg   # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0       # soak_main_kernels.py <end> [<first>]
for case in range(first, cases):
    rng = np.random.default_rng(9000 + case)
    F = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 12, 13, 16]))
    kw = dict(num_food_items=F, forced_breathing=bool(rng.random() < 0.7), random_food_count=bool(rng.random() < 0.3),
              respawn_food=bool(rng.random() < 0.75), proximity_reward_weight=float(rng.choice([0.0, 0.5, 5.0])),
              efficiency_bonus=float(rng.choice([0.0, 1.0])), max_steps_without_food=int(rng.integers(30, 500)),
              food_reward=float(rng.uniform(1, 20)), collision_penalty=float(-rng.uniform(1, 60)),
              time_penalty=float(-rng.uniform(0, 0.5)))
    other = case >= 30 and bool(rng.random() < 0.6)    # cases 30+: also the STD = false kernels (K = 3, another tank / physics)
    if other:
        kw.update(width=int(rng.integers(500, 1200)), height=int(rng.integers(450, 900)), tank_margin=float(rng.uniform(20, 60)),
                  base_radius=float(rng.uniform(18, 34)), max_thrust_force=float(rng.uniform(60, 160)),
                  drag_coefficient=float(rng.uniform(0.95, 0.995)), angular_drag=float(rng.uniform(0.9, 0.99)),
                  max_nozzle_angle=float(rng.uniform(0.6, 1.3)), nozzle_response_rate=float(rng.uniform(0.02, 0.2)),
                  food_radius=float(rng.uniform(8, 25)), min_food_distance=float(rng.uniform(40, 110)))
        if rng.random() < 0.5:
            kw.update(inhale_duration=int(rng.integers(10, 200)), exhale_duration=int(rng.integers(20, 250)),
                      rest_duration=int(rng.integers(0, 120)))
    cfg = pkg.load_env_config("single_food", **kw)
    n = 4096 + int(rng.integers(0, 200))
    H = 1100
    seed = int(rng.integers(0, 2 ** 31))
    want_final = bool(case & 1)
    act = T.make_actions(cfg, H, n, seed=case, scale=1.1)
    try:
        got, dev = T.run_device(cfg, n, act, seed=seed, want_final=want_final)
        orc = ol.OracleVec(cfg, n, seed=seed, threads=16)
        ref = orc.rollout(act, want_final=want_final)
        d = T.assert_parity(cfg, got, ref, f"case {case}")
        T.assert_state_parity(cfg, dev, orc, f"case {case}")
        st = dev.stats()
        ll = dev.last_launch()
        print(f"case {case}: F={F} literal={ll['literal_constants']} sig={ll['full_signature']} n={n} final={want_final} forced={kw['forced_breathing']} respawn={kw['respawn_food']} "
              f"max obs diff {d[0]:.2e} episodes {st['episodes']} food {st['food_collected']}", flush=True)
        dev.close(); orc.close()
    except AssertionError as e:
        bad += 1
        print("FAIL", case, kw, str(e)[:300], flush=True)
print("done, failures:", bad)
