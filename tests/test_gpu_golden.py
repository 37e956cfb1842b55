"""The HIP path (C ABI) against the golden vectors generated from the reference's own Python
implementation (tests/golden/ref_*.npz): flags and info integers identical, observations within
1e-5 (angle columns on the circle), rewards within 1e-5 * max(1, |r|), fp64 end state within 1e-9."""
import numpy as np
import pytest

from golden_util import fixture_names, load_fixture, obs_diff
from underwater_swimmer_rl_amd import _capi
from underwater_swimmer_rl_amd._capi import SalpLib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", fixture_names())
def test_hip_matches_reference_vectors(name):
    z, meta, cfg = load_fixture(name)
    act = np.ascontiguousarray(z["actions"])
    H, n, _ = act.shape
    dev = SalpLib(cfg, n, device_id=0, seed=meta["seed"], env_index_base=meta["env_index_base"])
    if "inject_f64" in z.files:
        dev.set_state(np.ascontiguousarray(z["inject_f64"]), np.ascontiguousarray(z["inject_i32"]), 0)
    obs0 = np.empty((n, cfg.obs_dim), np.float32)
    dev.observe(obs0, 0)
    assert obs_diff(cfg, obs0, z["reset_obs"]).max() <= 1e-5
    obs = np.empty((H, n, cfg.obs_dim), np.float32)
    fin = np.full((H, n, cfg.obs_dim), np.nan, np.float32)
    rew = np.empty((H, n), np.float32)
    term = np.empty((H, n), np.uint8)
    trunc = np.empty((H, n), np.uint8)
    info = np.empty((n, 3), np.int32)
    # step-by-step through salp_vec_step (info + final_obs), the reference's own call shape
    sched = {int(t): int(k) for t, k in z["food_schedule"]} if "food_schedule" in z.files else {}
    for t in range(H):
        if t in sched:          # the curriculum's base_num_food_items poke, through the C ABI
            dev.set_base_num_food(sched[t])
            assert dev.base_num_food == sched[t]
        dev.step(act[t], obs[t], rew[t], term[t], trunc[t], fin[t], info, 0)
        assert np.array_equal(info, z["info"][t]), f"info at step {t}"
    assert np.array_equal(term, z["terminated"]) and np.array_equal(trunc, z["truncated"])
    nan_ref = np.isnan(z["obs"])
    assert np.array_equal(np.isnan(obs), nan_ref)
    assert obs_diff(cfg, obs, z["obs"]).max() <= 1e-5
    done = (z["terminated"] | z["truncated"]).astype(bool)
    if cfg.no_autoreset:
        assert np.isnan(fin).all()                           # nothing is reset, so there is no terminal observation
        done[:] = False
    assert np.array_equal(~np.isnan(fin[..., 0]), done)      # terminal rows are written for finished envs only
    if done.any():
        assert obs_diff(cfg, fin[done], z["final_obs"][done]).max() <= 1e-5
    r = z["reward"]
    ok = np.isfinite(r)
    assert (np.abs(rew[ok] - r[ok]) / np.maximum(1.0, np.abs(r[ok]))).max() <= 1e-5
    f64 = np.empty((_capi.F_FOOD0 + 2 * cfg.num_food_items, n), np.float64)
    i32 = np.empty((_capi.I_COUNT, n), np.int32)
    dev.get_state(f64, i32, 0)
    assert np.array_equal(i32, z["end_i32"]), np.nonzero(i32 != z["end_i32"])
    e = z["end_f64"]
    assert np.array_equal(np.isnan(f64), np.isnan(e))
    assert np.nanmax(np.abs(f64 - e)) <= 1e-9
    dev.close()


@pytest.mark.parametrize("F", [4, 12, 16, "16_k5", "9_k5"])
def test_tie_order_in_whole_wavefronts_fused_rollout(F):
    """The tie-order vectors (tests/golden/gen_golden.py `tie_order_f*`: exact ties, squared distances a few ulp apart,
    a pair only fp64 can tell apart) through the UNPREDICATED fused rollout kernels: the five envs tiled 64 times into
    320 envs = five whole wavefronts, 130 steps in one launch (the swimmers rest until step ~135, so every copy sees the
    injected geometry; the copies differ from the vector only through their draw streams, which nothing uses before the
    first thrust).  F = 4 / 12 / 16: the K = 3 register-food kernels (12: sac_gail's); 16_k5 / 9_k5: five observed foods —
    the generic instantiations, foods in LDS (16) and in VGPRs (9)."""
    z, meta, cfg = load_fixture(f"tie_order_f{F}")
    reps, H = 64, 130
    n0 = z["actions"].shape[1]
    n = n0 * reps
    dev = SalpLib(cfg, n, device_id=0, seed=meta["seed"], env_index_base=meta["env_index_base"])
    dev.set_state(np.ascontiguousarray(np.tile(z["inject_f64"], (1, reps))), np.ascontiguousarray(np.tile(z["inject_i32"], (1, reps))), 0)
    act = np.ascontiguousarray(np.tile(z["actions"][:H], (1, reps, 1)))
    obs = np.empty((H, n, cfg.obs_dim), np.float32)
    rew = np.empty((H, n), np.float32)
    term = np.empty((H, n), np.uint8)
    trunc = np.empty((H, n), np.uint8)
    dev.rollout(act, H, obs, rew, term, trunc, None, None, 0)
    ll = dev.last_launch()
    assert ll["envs_unpredicated"] == n and ll["envs_predicated"] == 0
    assert ll["full_signature"] == (1 if isinstance(F, int) else 0)      # the generic (K != 3) instantiation tests every store
    assert (ll["food_slots"], ll["observed_capacity"]) == {4: (4, 3), 12: (12, 3), 16: (16, 3), "16_k5": (16, 8), "9_k5": (12, 8)}[F]
    want = np.tile(z["obs"][:H], (1, reps, 1))
    assert obs_diff(cfg, obs, want).max() <= 1e-5
    r = np.tile(z["reward"][:H], (1, reps))
    assert (np.abs(rew - r) / np.maximum(1.0, np.abs(r))).max() <= 1e-5
    assert not term.any() and not trunc.any()
    dev.close()
