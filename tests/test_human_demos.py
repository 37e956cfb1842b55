"""Known-answer data held by the reference itself: its recorded human demonstrations
(data/expert_demos/human/*.pkl -> tests/golden/human_demo_*.npz, extracted without unpickling by
tests/golden/extract_human_demos.py).  Replaying the recorded actions must reproduce the recorded base
observation: heading, angular velocity, body size, breathing phase, water volume and nozzle
(columns 4..9) are functions of the actions alone and must match BIT-FOR-BIT in the oracle;
position / velocity (columns 0..3) carry the un-recorded thrust jitter of legacy:311 and must
agree within 3e-5 (SURVEY.md §4 measured 2e-5).  Food is kept out of the way (num_food_items=0),
as the recorded food layout depends on the reference's un-seeded RNG."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
import underwater_swimmer_rl_amd as pkg

DEMOS = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "human_demo_*.npz")))


def test_demos_present():
    assert len(DEMOS) == 5


@pytest.mark.parametrize("path", DEMOS, ids=[os.path.basename(p)[:-4] for p in DEMOS])
def test_oracle_reproduces_recorded_human_demo(path):
    z = np.load(path, allow_pickle=False)
    act, rec = z["actions"], z["base_obs"]
    T = len(act)
    cfg = pkg.load_env_config("sac_gail", num_food_items=0)   # scripts/collection/collect_human_demos.py:276-285
    orc = ol.OracleVec(cfg, 1, seed=0)
    first = orc.observe()
    out = orc.rollout_f64(act.reshape(T, 1, 1))
    assert out["terminated"].sum() == 0
    obs = np.concatenate([first[None], out["obs"]])[:, 0, :10]
    assert np.array_equal(obs[:, 4:10], rec[:, 4:10]), "columns 4..9 must be bit-identical"
    assert np.abs(obs[:, :4] - rec[:, :4]).max() <= 3e-5


@pytest.mark.gpu
@pytest.mark.parametrize("path", DEMOS[:2], ids=[os.path.basename(p)[:-4] for p in DEMOS[:2]])
def test_hip_reproduces_recorded_human_demo(path):
    """Same replay through the C ABI (f32 actions, so the nozzle target carries an f32 rounding)."""
    from underwater_swimmer_rl_amd._capi import SalpLib
    z = np.load(path, allow_pickle=False)
    act, rec = z["actions"].astype(np.float32), z["base_obs"]
    T = len(act)
    cfg = pkg.load_env_config("sac_gail", num_food_items=0)
    dev = SalpLib(cfg, 1, device_id=0, seed=0)
    obs = np.empty((T, 1, cfg.obs_dim), np.float32)
    dev.rollout(np.ascontiguousarray(act.reshape(T, 1, 1)), T, obs, None, None, None, None, None, 0)
    got = obs[:, 0, :10]
    assert np.abs(got[:, 4:10] - rec[1:, 4:10]).max() <= 1e-5
    assert np.abs(got[:, :4] - rec[1:, :4]).max() <= 3e-5
    dev.close()
