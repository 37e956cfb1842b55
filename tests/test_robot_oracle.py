"""HEAD-simulator oracle (oracle/salp_robot_oracle.c) against vectors produced by the reference's own
robot.py + salp_robot_env.py (tests/golden/gen_robot_golden.py).  The reference's 3x3 products go
through numpy/BLAS whose summation order is unspecified, so the pin is a tolerance: flags and
inner-step counts identical; observations / rewards / fp64 end state within 1e-6 (relative to
max(1, |x|)) — the measured deviation is ~1e-8 after thousands of Euler steps."""
import glob
import json
import os

import numpy as np
import pytest

import robot_oracle_lib as rol

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "robot_*.npz")))
TOL = 1e-6


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.nanmax(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def test_fixtures_present():
    assert len(GOLD) == 3


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_robot_oracle_matches_reference_vectors(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    act = z["actions"]
    T, n, _ = act.shape
    orc = rol.RobotOracleVec(n, seed=meta["seed"], env_index_base=meta["env_index_base"])
    assert rel(orc.reset(np.zeros(n, np.uint8)), z["reset_obs"]) <= TOL
    for t in range(T):
        out = orc.step(act[t])
        assert np.array_equal(out["terminated"], z["terminated"][t]) and np.array_equal(out["truncated"], z["truncated"][t]), t
        assert np.array_equal(out["inner_steps"], z["inner_steps"][t]), t
        assert rel(out["obs"], z["obs"][t]) <= TOL and rel(out["reward"], z["reward"][t]) <= TOL, t
        done = (z["terminated"][t] | z["truncated"][t]).astype(bool)
        assert np.array_equal(~np.isnan(out["final_obs"][:, 0]), done)
        if done.any():
            assert rel(out["final_obs"][done], z["final_obs"][t][done]) <= TOL
    assert rel(orc.get_state(), z["end_state"]) <= TOL
    orc.close()


def test_cycle_structure():
    """One env step is a whole cycle: (contraction/0.02 + contraction/0.04 + coast) / 0.01 Euler steps."""
    orc = rol.RobotOracleVec(3, seed=1)
    a = np.array([[0.5, 0.1, 0.0], [0.0, 0.0, 0.3], [1.0, 1.0, -1.0]], np.float32)
    out = orc.step(a)
    expect = [(0.03 / 0.02 + 0.03 / 0.04 + 1.0) / 0.01, 0.0, (0.06 / 0.02 + 0.06 / 0.04 + 10.0) / 0.01]
    assert abs(out["inner_steps"][0] - expect[0]) <= 1 and out["inner_steps"][1] == 0 and abs(out["inner_steps"][2] - expect[2]) <= 1
    s = orc.get_state()
    assert s[rol.R_CYCLE].tolist() == [1.0, 1.0, 1.0]
    assert s[rol.R_POS + 2].max() < 1e-6          # planar motion: z stays ~0
    orc.close()
