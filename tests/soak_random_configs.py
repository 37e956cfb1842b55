"""Soak run (not collected by pytest): 120 further randomly drawn configurations, HIP path vs the oracle, with the
helpers of test_gpu_parity.py.  python3 tests/soak_random_configs.py   (last run: 0 failures)"""
import sys, numpy as np
import os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import test_gpu_parity as T
import oracle_lib as ol
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 220)
for case in range(lo, hi):
    rng = np.random.default_rng(5000 + case)
    cfg = T._random_cfg(rng)
    n, H, seed = 321, 200, int(rng.integers(0, 2 ** 31))
    act = T.make_actions(cfg, H, n, seed=case, scale=1.2)
    try:
        got, dev = T.run_device(cfg, n, act, seed=seed, want_final=True)
        orc = ol.OracleVec(cfg, n, seed=seed)
        ref = orc.rollout(act, want_final=True)
        T.assert_parity(cfg, got, ref, f"case {case}")
        T.assert_state_parity(cfg, dev, orc, f"case {case}")
        dev.close(); orc.close()
    except AssertionError as e:
        bad += 1
        print("FAIL", case, cfg, str(e)[:300], flush=True)
print("done, failures:", bad)
