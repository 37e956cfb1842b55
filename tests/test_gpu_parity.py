"""GPU parity tests: the HIP path (through the C ABI of include/salp_vec.h) against the CPU oracle
on the same seeded inputs.  Run on the MI355X box with `pytest -m gpu`.

Tolerances (north_star: "within 1e-5 fp32"):
  * terminated / truncated / info integers: identical;
  * observation: max |diff| <= 1e-5 (columns holding an angle/pi are compared on the circle,
    i.e. modulo 2, because -1 and +1 are the same heading — snake:403-407 wraps to [-pi, pi]);
  * reward: |diff| <= 1e-5 * max(1, |reward|);
  * fp64 state snapshot: |diff| <= 1e-9 (only device sin/cos and d^2-vs-sqrt predicates differ).
"""
import numpy as np
import pytest

import oracle_lib as ol
import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd import _capi
from underwater_swimmer_rl_amd._capi import SalpLib

pytestmark = pytest.mark.gpu

OBS_TOL = 1e-5
REW_TOL = 1e-5
STATE_TOL = 1e-9


def angle_cols(cfg):
    return [4] + [10 + 4 * s + 3 for s in range(cfg.max_observed_food)]


def obs_diff(cfg, a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    for c in angle_cols(cfg):
        d[..., c] = np.minimum(d[..., c], 2.0 - d[..., c])
    return d


def run_device(cfg, n, act=None, horizon=None, seed=0, base=0, want_final=False, dev=None):
    own = dev is None
    if own:
        dev = SalpLib(cfg, n, device_id=0, seed=seed, env_index_base=base)
    H = act.shape[0] if act is not None else horizon
    obs = np.empty((H, n, cfg.obs_dim), np.float32)
    rew = np.empty((H, n), np.float32)
    term = np.empty((H, n), np.uint8)
    trunc = np.empty((H, n), np.uint8)
    fin = np.full((H, n, cfg.obs_dim), np.nan, np.float32) if want_final else None
    aout = np.empty((H, n, cfg.act_dim), np.float32) if act is None else None
    dev.rollout(act, H, obs, rew, term, trunc, fin, aout, 0)
    out = dict(obs=obs, reward=rew, terminated=term, truncated=trunc, final_obs=fin, actions=aout)
    return (out, dev) if not own else (out, dev)


def get_state(dev, cfg):
    f64 = np.empty((_capi.F_FOOD0 + 2 * cfg.num_food_items, dev.n_envs), np.float64)
    i32 = np.empty((_capi.I_COUNT, dev.n_envs), np.int32)
    dev.get_state(f64, i32, 0)
    return f64, i32


def assert_parity(cfg, got, ref, label=""):
    assert np.array_equal(got["terminated"], ref["terminated"]), f"{label}: terminated flags differ"
    assert np.array_equal(got["truncated"], ref["truncated"]), f"{label}: truncated flags differ"
    d = obs_diff(cfg, got["obs"], ref["obs"])
    assert d.max() <= OBS_TOL, f"{label}: obs diff {d.max()} at {np.unravel_index(d.argmax(), d.shape)}"
    rd = np.abs(got["reward"].astype(np.float64) - ref["reward64"]) / np.maximum(1.0, np.abs(ref["reward64"]))
    assert rd.max() <= REW_TOL, f"{label}: reward diff {rd.max()}"
    return float(d.max()), float(rd.max())


def assert_state_parity(cfg, dev, orc, label=""):
    f_d, i_d = get_state(dev, cfg)
    f_o, i_o = orc.get_state()
    assert np.array_equal(i_d, i_o), f"{label}: integer state differs in rows {np.unique(np.nonzero(i_d != i_o)[0])}"
    both_nan = np.isnan(f_d) & np.isnan(f_o)
    assert np.array_equal(np.isnan(f_d), np.isnan(f_o)), f"{label}: food None-pattern differs"
    d = np.where(both_nan, 0.0, np.abs(f_d - f_o))
    assert d.max() <= STATE_TOL, f"{label}: fp64 state diff {d.max()} in row {np.unravel_index(d.argmax(), d.shape)}"


CASES = {
    "single_food": dict(preset="single_food"),
    "long_horizon": dict(preset="single_food_long_horizon"),
    "sac_gail_F12": dict(preset="sac_gail"),
    "free_breathing": dict(preset="single_food", forced_breathing=False),
    "no_respawn_F3": dict(preset="sac_gail", num_food_items=3, respawn_food=False),
    "random_count_F5": dict(preset="sac_gail", num_food_items=5, random_food_count=True),
    "class_default_F5": dict(preset="sac_gail", num_food_items=5),          # the 8-slot register-food instantiation
    "F8_all_slots": dict(preset="sac_gail", num_food_items=8, max_steps_without_food=150),
    # K = 3 with non-default constants: the per-slot-count instantiations that read their constants from the launch parameters
    "other_tank_F1": dict(preset="single_food", width=900, height=700, tank_margin=40.0),
    "other_physics_F12": dict(preset="sac_gail", drag_coefficient=0.97, max_thrust_force=120.0, base_radius=26.0,
                              inhale_duration=100, exhale_duration=130, nozzle_response_rate=0.08),
    "other_tank_F5_free": dict(preset="sac_gail", num_food_items=5, width=1000, forced_breathing=False, min_food_distance=60.0),
    "K2_generic": dict(preset="sac_gail", num_food_items=6, max_observed_food=2, proximity_reward_weight=2.0),
    "K0_no_food_obs": dict(preset="single_food", max_observed_food=0),
    "F0_empty": dict(preset="single_food", num_food_items=0),
    "short_timeout": dict(preset="single_food", max_steps_without_food=40),
    # the unpredicated (main-launch) forms of the instantiations that only ran predicated before round 3:
    "F16_sixteen_slots": dict(preset="sac_gail", num_food_items=16, max_steps_without_food=200),        # <16, 3, STD>
    "F16_sixteen_slots_other_tank": dict(preset="sac_gail", num_food_items=16, width=900, height=650),  # <16, 3, !STD>
    "F14_K5_generic_lds": dict(preset="sac_gail", num_food_items=14, max_observed_food=5),              # <16, 8>: generic, foods in LDS
    "F9_K5_generic_reg": dict(preset="sac_gail", num_food_items=9, max_observed_food=5),                # <12, 8>: generic, foods in VGPRs
    "F3_other_tank": dict(preset="sac_gail", num_food_items=3, width=900, tank_margin=40.0),            # <4, 3, !STD>
}
# (food slots, observed capacity, literal constants) of the kernel each case must run (salp_vec_last_launch)
EXPECT_KERNEL = {
    "single_food": (1, 3, 1), "sac_gail_F12": (12, 3, 1), "class_default_F5": (8, 3, 1), "no_respawn_F3": (4, 3, 1),
    "other_tank_F1": (1, 3, 0), "other_physics_F12": (12, 3, 0), "K2_generic": (12, 8, 0),
    "F16_sixteen_slots": (16, 3, 1), "F16_sixteen_slots_other_tank": (16, 3, 0), "F14_K5_generic_lds": (16, 8, 0),
    "F9_K5_generic_reg": (12, 8, 0), "F3_other_tank": (4, 3, 0),
}


def make_cfg(spec):
    spec = dict(spec)
    return pkg.load_env_config(spec.pop("preset"), **spec)


def make_actions(cfg, H, n, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    act = rng.uniform(-scale, scale, size=(H, n, cfg.act_dim)).astype(np.float32)
    if not cfg.forced_breathing:  # inhale control in [0,1], held for random stretches
        hold = rng.uniform(0, 1, size=(H // 16 + 1, n)).repeat(16, axis=0)[:H]
        act[..., 0] = hold.astype(np.float32)
    return act


@pytest.mark.parametrize("name", list(CASES))
def test_rollout_parity(name):
    cfg = make_cfg(CASES[name])
    n, H, seed = 2048, 384, 11
    act = make_actions(cfg, H, n, seed=3)
    got, dev = run_device(cfg, n, act, seed=seed, want_final=True)
    orc = ol.OracleVec(cfg, n, seed=seed)
    ref = orc.rollout(act, want_final=True)
    dmax, rmax = assert_parity(cfg, got, ref, name)
    # terminal observations are delivered for finished envs only
    done = (ref["terminated"] | ref["truncated"]).astype(bool)
    assert np.array_equal(np.isnan(got["final_obs"][..., 0]), ~done)
    if done.any():
        assert obs_diff(cfg, got["final_obs"][done], ref["final_obs"][done]).max() <= OBS_TOL
    assert_state_parity(cfg, dev, orc, name)
    st = dev.stats()
    assert st["env_steps"] == n * H
    assert st["episodes"] == int(done.sum())
    assert st["terminated"] == int(ref["terminated"].sum()) and st["truncated"] == int(ref["truncated"].sum())
    ll = dev.last_launch()      # 2048 envs = 32 whole wavefronts: the unpredicated kernel, non-FULL signature (final_obs)
    assert ll["envs_unpredicated"] == n and ll["envs_predicated"] == 0
    assert ll["full_signature"] == (2 if ll["observed_capacity"] == 3 else 0)    # main outputs + final_obs + info
    if name in EXPECT_KERNEL:
        assert (ll["food_slots"], ll["observed_capacity"], ll["literal_constants"]) == EXPECT_KERNEL[name], ll
    print(f"{name}: max obs diff {dmax:.3g}, max rel reward diff {rmax:.3g}, episodes {st['episodes']}, kernel {ll}")
    dev.close()


@pytest.mark.parametrize("foods,slots", [(3, 4), (5, 8), (12, 12), (16, 16)])
def test_multi_food_full_signature_main_launch(foods, slots):
    """The FULL-signature unpredicated kernels (what bench.py and salp_vec_rollout without final_obs run) of every
    multi-food slot count, with captures, respawns and truncation resets inside whole wavefronts."""
    cfg = pkg.load_env_config("sac_gail", num_food_items=foods, max_steps_without_food=120)
    n, H, seed = 1024, 400, 31 + foods
    act = make_actions(cfg, H, n, seed=foods)
    got, dev = run_device(cfg, n, act, seed=seed)
    orc = ol.OracleVec(cfg, n, seed=seed)
    ref = orc.rollout(act)
    assert_parity(cfg, got, ref, f"F{foods} full signature")
    assert_state_parity(cfg, dev, orc, f"F{foods} full signature")
    ll = dev.last_launch()
    assert ll["food_slots"] == slots and ll["observed_capacity"] == 3 and ll["literal_constants"] == 1
    assert ll["full_signature"] == 1 and ll["envs_unpredicated"] == n and ll["envs_predicated"] == 0
    assert dev.stats()["truncated"] > 100
    dev.close()


def test_config2_4096x256_single_food():
    """BASELINE.json configs[1]: N_envs=4096 single_food, 256-step rollout, fp32 diff <= 1e-5."""
    cfg = pkg.load_env_config("single_food")
    n, H = 4096, 256
    act = make_actions(cfg, H, n, seed=0)
    got, dev = run_device(cfg, n, act, seed=0)
    ref = ol.OracleVec(cfg, n, seed=0).rollout(act)
    dmax, rmax = assert_parity(cfg, got, ref, "config2")
    print(f"config2: max obs diff {dmax:.3g}, max rel reward diff {rmax:.3g}")
    dev.close()


def test_long_rollout_wall_and_food_events():
    """2000 steps: many wall terminations, food captures + respawns, the rounding-escape case."""
    cfg = pkg.load_env_config("single_food_long_horizon")
    n, H = 512, 2000
    act = make_actions(cfg, H, n, seed=9)
    got, dev = run_device(cfg, n, act, seed=5)
    orc = ol.OracleVec(cfg, n, seed=5)
    ref = orc.rollout(act)
    assert ref["terminated"].sum() > 100
    assert_parity(cfg, got, ref, "long")
    assert_state_parity(cfg, dev, orc, "long")
    dev.close()


def test_step_equals_rollout_and_info():
    cfg = pkg.load_env_config("sac_gail")
    n, H, seed = 777, 300, 21   # ragged: not a multiple of 64
    act = make_actions(cfg, H, n, seed=4)
    got, dev_r = run_device(cfg, n, act, seed=seed)
    dev_s = SalpLib(cfg, n, device_id=0, seed=seed)
    orc = ol.OracleVec(cfg, n, seed=seed)
    ref = orc.rollout(act)
    obs = np.empty((n, cfg.obs_dim), np.float32)
    rew = np.empty(n, np.float32)
    term = np.empty(n, np.uint8)
    trunc = np.empty(n, np.uint8)
    info = np.empty((n, 3), np.int32)
    for t in range(H):
        dev_s.step(act[t], obs, rew, term, trunc, None, info, 0)
        assert np.array_equal(obs, got["obs"][t]) and np.array_equal(rew, got["reward"][t])
        assert np.array_equal(term, got["terminated"][t]) and np.array_equal(trunc, got["truncated"][t])
        assert np.array_equal(info, ref["info"][t]), f"info differs at step {t}"
    dev_r.close()
    dev_s.close()


def test_reset_observation_and_mask():
    cfg = pkg.load_env_config("sac_gail")
    n, seed = 1000, 3
    dev = SalpLib(cfg, n, device_id=0, seed=seed)
    orc = ol.OracleVec(cfg, n, seed=seed)
    obs = np.empty((n, cfg.obs_dim), np.float32)
    dev.observe(obs, 0)
    assert obs_diff(cfg, obs, orc.observe()).max() <= OBS_TOL
    act = make_actions(cfg, 50, n, seed=1)
    run_device(cfg, n, act, dev=dev)
    orc.rollout(act)
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    dev.reset(mask, obs, 0)
    ref = orc.reset(mask)
    assert obs_diff(cfg, obs, ref).max() <= OBS_TOL
    assert_state_parity(cfg, dev, orc, "masked reset")
    dev.close()


@pytest.mark.parametrize("n,preset,over", [(640, "single_food", {}), (700, "sac_gail", {}),
                                            (333, "single_food", dict(forced_breathing=False))])
def test_device_generated_actions_match_oracle_stream(n, preset, over):
    """act == NULL: in-kernel generation (FULL signature), full and ragged wavefronts, 1 and 2 actions."""
    cfg = pkg.load_env_config(preset, **over)
    H, seed = 203, 99
    got, dev = run_device(cfg, n, None, horizon=H, seed=seed)
    orc = ol.OracleVec(cfg, n, seed=seed)
    ref = orc.rollout(None, horizon=H)
    assert np.array_equal(got["actions"], ref["actions"])
    assert got["actions"][..., -1].min() >= -1.0 and got["actions"][..., -1].max() < 1.0
    assert_parity(cfg, got, ref, "device actions")
    # second launch continues the action stream at global step H
    got2, _ = run_device(cfg, n, None, horizon=50, dev=dev)
    ref2 = orc.rollout(None, horizon=50)
    assert np.array_equal(got2["actions"], ref2["actions"])
    assert_parity(cfg, got2, ref2, "device actions, 2nd launch")
    dev.close()


def test_set_state_injection_eval_style():
    """eval/collect_navigation_data.py:76-89: overwrite pose, velocity, heading and the food."""
    cfg = pkg.load_env_config("single_food", respawn_food=False, max_steps_without_food=3000)
    n, seed = 256, 1
    dev = SalpLib(cfg, n, device_id=0, seed=seed)
    orc = ol.OracleVec(cfg, n, seed=seed)
    f64, i32 = get_state(dev, cfg)
    rng = np.random.default_rng(2)
    f64[_capi.F_X] = 150.0
    f64[_capi.F_Y] = 300.0
    f64[_capi.F_VX] = 0.0
    f64[_capi.F_VY] = 0.0
    f64[_capi.F_THETA] = rng.uniform(-np.pi, np.pi, n)
    f64[_capi.F_OMEGA] = 0.0
    f64[_capi.F_FOOD0] = 650.0
    f64[_capi.F_FOOD0 + 1] = 300.0
    i32[_capi.I_STEPS_SINCE_FOOD] = 0
    dev.set_state(f64, i32, 0)
    fo, io_ = orc.get_state()
    fo[:] = f64
    fo[_capi.F_ELLIPSE_A] = 30.0
    fo[_capi.F_ELLIPSE_B] = 30.0
    orc.set_state(fo, i32)
    obs = np.empty((n, cfg.obs_dim), np.float32)
    dev.observe(obs, 0)
    assert obs_diff(cfg, obs, orc.observe()).max() <= OBS_TOL
    act = make_actions(cfg, 600, n, seed=8, scale=0.3)
    got, _ = run_device(cfg, n, act, dev=dev)
    ref = orc.rollout(act)
    assert_parity(cfg, got, ref, "injected")
    assert_state_parity(cfg, dev, orc, "injected")
    dev.close()


def test_out_of_range_and_nan_actions():
    """Actions are not clipped by the reference (legacy:125-135); NaN ends up at +max nozzle."""
    cfg = pkg.load_env_config("single_food")
    n, H = 128, 120
    act = make_actions(cfg, H, n, seed=6, scale=3.0)
    act[5::17, 3::7, 0] = np.nan
    got, dev = run_device(cfg, n, act, seed=2)
    orc = ol.OracleVec(cfg, n, seed=2)
    ref = orc.rollout(act)
    assert np.array_equal(got["terminated"], ref["terminated"])
    ok = ~np.isnan(ref["obs"]).any(axis=-1)
    assert obs_diff(cfg, got["obs"][ok], ref["obs"][ok]).max() <= OBS_TOL
    assert np.array_equal(np.isnan(got["obs"]), np.isnan(ref["obs"]))
    dev.close()


def test_sharding_is_trajectory_invariant():
    """env i's trajectory depends on its GLOBAL index only (multi-GPU sharding by env index)."""
    cfg = pkg.load_env_config("sac_gail")
    n, H, seed = 512, 150, 17
    act = make_actions(cfg, H, n, seed=12)
    full, d0 = run_device(cfg, n, act, seed=seed)
    lo, d1 = run_device(cfg, n // 2, np.ascontiguousarray(act[:, : n // 2]), seed=seed, base=0)
    hi, d2 = run_device(cfg, n // 2, np.ascontiguousarray(act[:, n // 2:]), seed=seed, base=n // 2)
    for k in ("obs", "reward", "terminated", "truncated"):
        assert np.array_equal(full[k][:, : n // 2], lo[k]) and np.array_equal(full[k][:, n // 2:], hi[k]), k
    for d in (d0, d1, d2):
        d.close()


@pytest.mark.parametrize("foods", [1, 5, 12, 16])
def test_partial_output_signatures_match_oracle(foods):
    """Callers may leave any output NULL (include/salp_vec.h): the kernels compiled for that case test every store.  Two
    such signatures per slot count in the unpredicated launch (2048 envs) — no reward / truncated, and reward + flags
    without observations (with terminal observations) — against the oracle, and the state after both."""
    cfg = make_cfg(dict(preset="sac_gail", num_food_items=foods, max_steps_without_food=60))
    n, H, seed = 2048, 200, 5
    act = make_actions(cfg, 2 * H, n, seed=9)
    dev = SalpLib(cfg, n, device_id=0, seed=seed)
    orc = ol.OracleVec(cfg, n, seed=seed)
    ref1 = orc.rollout(act[:H], want_final=True)
    ref2 = orc.rollout(act[H:], want_final=True)
    obs = np.empty((H, n, cfg.obs_dim), np.float32)
    term = np.empty((H, n), np.uint8)
    dev.rollout(np.ascontiguousarray(act[:H]), H, obs, None, term, None, None, None, 0)
    ll = dev.last_launch()
    assert ll["full_signature"] == 0 and ll["envs_unpredicated"] == n and ll["food_slots"] >= foods
    assert np.array_equal(term, ref1["terminated"])
    assert obs_diff(cfg, obs, ref1["obs"]).max() <= OBS_TOL
    rew = np.empty((H, n), np.float32)
    trunc = np.empty((H, n), np.uint8)
    fin = np.full((H, n, cfg.obs_dim), np.nan, np.float32)
    dev.rollout(np.ascontiguousarray(act[H:]), H, None, rew, term, trunc, fin, None, 0)
    assert dev.last_launch()["full_signature"] == 0
    assert np.array_equal(term, ref2["terminated"]) and np.array_equal(trunc, ref2["truncated"])
    r = ref2["reward"]
    assert (np.abs(rew - r) / np.maximum(1.0, np.abs(r))).max() <= REW_TOL
    done = (ref2["terminated"] | ref2["truncated"]).astype(bool)
    assert done.any() and np.array_equal(np.isnan(fin[..., 0]), ~done)
    assert obs_diff(cfg, fin[done], ref2["final_obs"][done]).max() <= OBS_TOL
    assert_state_parity(cfg, dev, orc, f"partial_F{foods}")
    dev.close()
    orc.close()


@pytest.mark.parametrize("foods,tank,min_wg,max_vgprs", [
    (1, False, 4, 128), (3, False, 4, 128), (5, False, 4, 128), (8, False, 4, 128), (12, False, 3, 168), (16, False, 3, 168),
    (5, True, 4, 128), (8, True, 4, 128), (1, True, 4, 128), (12, True, 3, 168), (16, True, 2, 256)])
def test_kernel_occupancy_matches_the_design(foods, tank, min_wg, max_vgprs):
    """DESIGN.md section 3.1's occupancy table as the runtime reports it for the kernels actually launched (no timing):
    workgroups of 256 threads resident per CU (= wavefronts per SIMD), VGPRs, no scratch — for the main-only signature and
    for the one with terminal observations, literal constants and an 801-wide tank.  (Two signatures of the 4- / 8-slot kernels
    once sat at 129 VGPRs — three per SIMD, 22 % slower — without any test noticing.)"""
    cfg = make_cfg(dict(preset="sac_gail", num_food_items=foods, **(dict(width=801) if tank else {})))
    n, H = 2048, 4
    act = make_actions(cfg, H, n, seed=1)
    dev = SalpLib(cfg, n, device_id=0, seed=3)
    for want_final in (False, True):
        run_device(cfg, n, act, dev=dev, want_final=want_final)
        ll, res = dev.last_launch(), dev.last_kernel_resources()
        assert ll["literal_constants"] == (0 if tank else 1) and ll["full_signature"] == (2 if want_final else 1)
        assert res["workgroups_per_cu"] >= min_wg, (ll, res)
        assert res["vgprs"] <= max_vgprs and res["scratch_bytes"] <= (32 if tank else 0), (ll, res)   # (801-wide, 12 slots: 2 registers)
    dev.close()


def test_food_next_to_the_swimmer_keeps_the_reward_exact():
    """A fallback placement (snake:120-131, :270-276) can leave a food next to the swimmer; with two foods inside the capture
    radius the one in the later slot survives a step, and the shaped reward w cos(bearing) is then taken on an offset of a
    fraction of a pixel.  In fp32 (roundings of both positions, ~6e-5 px) the bearing of a food 0.7 px away was off by
    1.7e-5 rad, the reward of w = 5 by 8e-5 (tests/soak_main_kernels.py case 92); the nearest food's offsets now come from the
    exact positions when it is that close.  Injected: 12 foods, two of them 0.4 and 0.9 px from the resting swimmer."""
    cfg = make_cfg(dict(preset="sac_gail", proximity_reward_weight=5.0))
    n, H = 128, 3
    dev = SalpLib(cfg, n, device_id=0, seed=21)
    orc = ol.OracleVec(cfg, n, seed=21)
    f64, i32 = get_state(dev, cfg)
    rng = np.random.default_rng(4)
    ang = rng.uniform(-np.pi, np.pi, size=(2, n))
    F = cfg.num_food_items
    for j, (slot, d) in enumerate(((7, 0.4), (3, 0.9))):     # the nearer one sits in the LATER slot: it survives the first step
        f64[_capi.F_FOOD0 + slot] = f64[_capi.F_X] + d * np.cos(ang[j])
        f64[_capi.F_FOOD0 + F + slot] = f64[_capi.F_Y] + d * np.sin(ang[j])
    dev.set_state(f64, i32, 0)
    orc.set_state(f64, i32)
    act = np.zeros((H, n, cfg.act_dim), np.float32)
    got, _ = run_device(cfg, n, act, dev=dev)
    ref = orc.rollout(act)
    assert ref["reward"][0].max() > 10 and ref["reward"][1].max() > 10      # a capture in each of the first two steps
    assert_parity(cfg, got, ref, "near food")
    r = ref["reward"]
    assert (np.abs(got["reward"] - r) / np.maximum(1.0, np.abs(r))).max() <= 3e-6
    dev.close()
    orc.close()


def test_properties_full_size_262144():
    """BASELINE.json configs[2] size, checked through size-independent properties: bounded
    observations, breathing period 273 in forced mode, |nozzle| <= 1, positions inside the tank,
    env-step accounting, and agreement with the oracle on a strided sample of 256 envs."""
    cfg = pkg.load_env_config("single_food_long_horizon")
    n, H, seed = 262144, 300, 0
    dev = SalpLib(cfg, n, device_id=0, seed=seed)
    obs = np.empty((H, n, cfg.obs_dim), np.float32)
    term = np.empty((H, n), np.uint8)
    aout = np.empty((H, n, 1), np.float32)
    dev.rollout(None, H, obs, None, term, None, None, aout, 0)
    assert np.isfinite(obs).all()
    assert obs[..., 0].min() >= (50 + 24) / 800 - 1e-6 and obs[..., 0].max() <= (750 - 24) / 800 + 1e-6
    assert obs[..., 1].min() >= (50 + 24) / 600 - 1e-6 and obs[..., 1].max() <= (550 - 24) / 600 + 1e-6
    assert np.abs(obs[..., 9]).max() <= 1.0 + 1e-6
    assert obs[..., 8].min() >= 0.0 and obs[..., 8].max() <= 1.0
    assert np.abs(obs[..., 4]).max() <= 1.0 + 1e-6
    alive = term[:284].sum(axis=0) == 0                     # envs that did not reset in the first cycle
    assert alive.sum() > n // 2
    assert np.array_equal(obs[0, alive, 6], obs[273, alive, 6])   # body size repeats with period 273
    assert np.array_equal(obs[10, alive, 7], obs[283, alive, 7])
    assert dev.stats()["env_steps"] == n * H
    idx = np.arange(0, n, n // 256)[:256]
    for j, i in enumerate(idx[:64]):
        orc = ol.OracleVec(cfg, 1, seed=seed, env_index_base=int(i))
        ref = orc.rollout(np.ascontiguousarray(aout[:, i:i + 1]))
        assert np.array_equal(term[:, i], ref["terminated"][:, 0])
        assert obs_diff(cfg, obs[:, i], ref["obs"][:, 0]).max() <= OBS_TOL
    dev.close()


def test_create_rejects_bad_arguments():
    lib = _capi.load_library()
    cfg = pkg.load_env_config("single_food")
    with pytest.raises(_capi.SalpError):
        SalpLib(cfg, 0)
    with pytest.raises(_capi.SalpError):
        SalpLib(cfg, 16, device_id=99)
    c = cfg.to_c()
    c.struct_size = 8
    import ctypes
    h = ctypes.c_void_p()
    assert lib.salp_vec_create(ctypes.byref(c), 16, 0, 0, 0, ctypes.byref(h)) == -1
    assert b"struct_size" in lib.salp_last_error()


def _random_cfg(rng):
    """A random but valid environment: food slots 0..16, observed foods 0..8, every flag, non-default tank,
    radii, drag, thrust and durations (the generic <16, 8> instantiation and the LDS food path), or the
    reference's constants with random flags (the STD instantiations)."""
    std = rng.random() < 0.4
    kw = dict(num_food_items=int(rng.integers(0, 17)), max_observed_food=int(rng.integers(0, 9)) if not std else 3,
              forced_breathing=bool(rng.random() < 0.6), random_food_count=bool(rng.random() < 0.3),
              respawn_food=bool(rng.random() < 0.7), proximity_reward_weight=float(rng.choice([0.0, 0.5, 5.0])),
              efficiency_bonus=float(rng.choice([0.0, 1.0])), max_steps_without_food=int(rng.integers(20, 400)),
              food_reward=float(rng.uniform(1, 20)), collision_penalty=float(-rng.uniform(1, 60)),
              time_penalty=float(-rng.uniform(0, 0.5)))
    if not std:
        kw.update(width=int(rng.integers(500, 1200)), height=int(rng.integers(450, 900)),
                  tank_margin=float(rng.uniform(20, 60)), base_radius=float(rng.uniform(18, 34)),
                  max_thrust_force=float(rng.uniform(60, 160)), drag_coefficient=float(rng.uniform(0.95, 0.995)),
                  angular_drag=float(rng.uniform(0.9, 0.99)), max_nozzle_angle=float(rng.uniform(0.6, 1.3)),
                  nozzle_response_rate=float(rng.uniform(0.02, 0.2)), food_radius=float(rng.uniform(8, 25)),
                  min_food_distance=float(rng.uniform(40, 110)))
        if rng.random() < 0.5:      # other breathing timings than the legacy 120 / 150 / 60
            kw.update(inhale_duration=int(rng.integers(10, 200)), exhale_duration=int(rng.integers(20, 250)),
                      rest_duration=int(rng.integers(0, 120)))
    return pkg.SalpSnakeConfig(**kw)


@pytest.mark.parametrize("case", range(24))
def test_random_configuration_parity(case):
    rng = np.random.default_rng(1000 + case)
    cfg = _random_cfg(rng)
    # 20 cases of 777 envs: a small ragged batch (n H <= 2^22) runs ONE predicated launch over the whole range
    # (launch_rollout), so these exercise the RAGGED = true kernels only, with final_obs (non-FULL signature).
    # Cases 5, 11, 17, 23: 16449 envs x 260 steps (n H > 2^22) split into the unpredicated main launch over 257 whole
    # wavefronts plus a one-env predicated launch, FULL signature.
    big = case % 6 == 5
    n, H, seed = (16449 if big else 777), 260, int(rng.integers(0, 2 ** 31))
    act = make_actions(cfg, H, n, seed=case, scale=1.2)            # a little outside the Box too
    got, dev = run_device(cfg, n, act, seed=seed, want_final=not big)
    orc = ol.OracleVec(cfg, n, seed=seed, threads=8 if big else 1)
    ref = orc.rollout(act, want_final=not big)
    assert_parity(cfg, got, ref, f"random case {case}: {cfg}")
    assert_state_parity(cfg, dev, orc, f"random case {case}")
    ll = dev.last_launch()
    assert (ll["envs_unpredicated"], ll["envs_predicated"]) == ((16448, 1) if big else (0, 777)), ll
    if not big:
        done = (ref["terminated"] | ref["truncated"]).astype(bool)
        if done.any():
            assert obs_diff(cfg, got["final_obs"][done], ref["final_obs"][done]).max() <= OBS_TOL
    dev.close()
    orc.close()


@pytest.mark.parametrize("foods", [4, 12, 16])
def test_near_tie_food_order_matches_oracle_across_scales(foods):
    """The order of the observed foods when two of them are ALMOST equally far — relative distance gaps from 1e-16 to
    1e-4, at distances 60..200 px — must be the reference's (stable sort on the fp64 distance,
    snake:382).  The fp32 ordering pass (csrc/salp_food_reg.h) may only decide where its error bound separates the keys
    and must fall back to the exact order otherwise (4, 12 and 16 slots).  A wrong
    bound would show as two swapped food blocks (bearing columns differ by O(1)).  4096 resting swimmers (zero velocity:
    the geometry holds for the whole rollout), observe() and 20 fused steps."""
    cfg = pkg.load_env_config("sac_gail", num_food_items=foods, proximity_reward_weight=1.0)
    n, H, seed = 4096, 20, 77
    dev = SalpLib(cfg, n, device_id=0, seed=seed)
    orc = ol.OracleVec(cfg, n, seed=seed, threads=8)
    f64, i32 = get_state(dev, cfg)
    rng = np.random.default_rng(5)
    x = rng.uniform(380, 420, n); y = rng.uniform(280, 320, n)
    f64[_capi.F_X], f64[_capi.F_Y] = x, y
    f64[_capi.F_VX] = f64[_capi.F_VY] = f64[_capi.F_OMEGA] = 0.0
    f64[_capi.F_THETA] = rng.uniform(-3, 3, n)
    F = foods
    # every food far away first, all distinct
    for k in range(F):
        ang = rng.uniform(0, 2 * np.pi, n)
        rad = rng.uniform(215, 235, n) + 0.37 * k
        f64[_capi.F_FOOD0 + k] = x + rad * np.cos(ang)
        f64[_capi.F_FOOD0 + F + k] = y + 0.85 * rad * np.sin(ang)
    # two (sometimes three) slots, in random slot order, at nearly the same distance d in [45, 200]
    d = rng.uniform(60, 200, n)                       # outside the capture radius (<= 54 px)
    gap = 10.0 ** rng.uniform(-16, -4, n) * rng.choice([-1.0, 1.0], n)
    slots = np.argsort(rng.random((n, F)), axis=1)[:, :3]
    angs = rng.uniform(0, 2 * np.pi, (n, 3))
    dist = np.stack([d, d * (1.0 + gap), np.where(rng.random(n) < 0.3, d * (1.0 - 0.5 * gap), d + 60.0)], axis=1)
    for j in range(3):
        f64[_capi.F_FOOD0 + slots[:, j], np.arange(n)] = x + dist[:, j] * np.cos(angs[:, j])
        f64[_capi.F_FOOD0 + F + slots[:, j], np.arange(n)] = y + dist[:, j] * np.sin(angs[:, j])
    i32[_capi.I_STEPS_SINCE_FOOD] = 0
    dev.set_state(f64, i32, 0)
    fo, _ = orc.get_state()
    fo[:] = f64
    fo[_capi.F_ELLIPSE_A] = 30.0
    fo[_capi.F_ELLIPSE_B] = 30.0
    orc.set_state(fo, i32)
    obs = np.empty((n, cfg.obs_dim), np.float32)
    dev.observe(obs, 0)
    d0 = obs_diff(cfg, obs, orc.observe())
    assert d0.max() <= OBS_TOL, f"observe(): {d0.max()} at {np.unravel_index(d0.argmax(), d0.shape)}"
    act = np.zeros((H, n, 1), np.float32)
    got, _ = run_device(cfg, n, act, dev=dev)
    ref = orc.rollout(act)
    assert_parity(cfg, got, ref, f"near ties, {foods} foods")
    assert dev.last_launch()["food_slots"] == {4: 4, 12: 12, 16: 16}[foods]
    dev.close()
    orc.close()
