"""HEAD simulator on the GPU: the HIP kernel through the C ABI of include/salp_robot.h
against (a) the vectors produced by the reference's own robot.py / salp_robot_env.py and (b) the C oracle
on seeded batches.  Flags and Euler-step counts identical; observations / rewards / fp64 state within
1e-6 relative to max(1, |x|) (the reference's 3x3 products go through BLAS, see test_robot_oracle.py)."""
import glob
import json
import os

import numpy as np
import pytest

import robot_oracle_lib as rol
from underwater_swimmer_rl_amd.robot_env import SalpRobotVectorEnv

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "robot_*.npz")))
TOL = 1e-6


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.nanmax(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_hip_robot_matches_reference_vectors(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    act = z["actions"]
    T, n, _ = act.shape
    env = SalpRobotVectorEnv(n, device=0, seed=meta["seed"], env_index_base=meta["env_index_base"], output="numpy")
    assert rel(env.observe(), z["reset_obs"]) <= TOL
    for t in range(T):
        obs, rew, term, trunc, info = env.step(act[t])
        assert np.array_equal(term, z["terminated"][t].astype(bool)) and np.array_equal(trunc, z["truncated"][t].astype(bool)), t
        assert np.array_equal(info["inner_steps"], z["inner_steps"][t]), t
        assert rel(obs, z["obs"][t]) <= TOL and rel(rew, z["reward"][t]) <= 1e-5, t
        done = term | trunc
        if done.any():
            assert rel(info["final_observation"][done], z["final_obs"][t][done]) <= TOL
    assert rel(env.get_state(), z["end_state"]) <= TOL
    env.close()


def test_hip_robot_matches_oracle_on_a_batch():
    n, seed, T = 3000, 5, 12          # ragged last wavefront; every lane its own cycle length
    env = SalpRobotVectorEnv(n, device="cuda:0", seed=seed)
    orc = rol.RobotOracleVec(n, seed=seed)
    rng = np.random.default_rng(2)
    assert rel(env.observe().cpu().numpy(), orc.reset(np.zeros(n, np.uint8))) <= TOL
    for t in range(T):
        a = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 0.15, n), rng.uniform(-1, 1, n)], axis=1).astype(np.float32)
        obs, rew, term, trunc, info = env.step(a)
        ref = orc.step(a)
        assert np.array_equal(info["inner_steps"].cpu().numpy(), ref["inner_steps"])
        assert np.array_equal(term.cpu().numpy(), ref["terminated"].astype(bool))
        assert rel(obs.cpu().numpy(), ref["obs"]) <= TOL and rel(rew.cpu().numpy(), ref["reward"]) <= 1e-5
    assert rel(env.get_state(), orc.get_state()) <= TOL
    env.close()
    orc.close()


def test_out_of_box_and_non_finite_actions_return_promptly():
    """An unsquashed policy output must not spin the kernel: a cycle longer than the action Box allows is cut at
    14.6 s (<= 1460 Euler steps of 0.01 s), a non-finite one runs no Euler step (include/salp_robot.h)."""
    import time
    import torch
    n = 256
    env = SalpRobotVectorEnv(n, device="cuda:0", seed=3)
    a = np.tile(np.array([0.5, 0.1, 0.0], np.float32), (n, 1))
    a[0] = (1.0e9, 1.0e9, 0.0)
    a[1] = (0.5, np.inf, 0.0)
    a[2] = (np.inf, 0.0, 0.0)
    a[3] = (np.nan, np.nan, np.nan)
    a[4] = (0.5, -np.inf, 0.0)
    t0 = time.time()
    obs, rew, term, trunc, info = env.step(a)
    torch.cuda.synchronize()
    assert time.time() - t0 < 5.0
    steps = info["inner_steps"].cpu().numpy()
    assert steps.max() <= 1461 and steps[0] >= 1459 and steps[1] >= 1459
    assert steps[3] == 0 and steps[4] == 0
    assert np.isfinite(obs.cpu().numpy()[5:]).all()
    obs, rew, term, trunc, info = env.step(np.tile(np.array([0.5, 0.1, 0.0], np.float32), (n, 1)))   # still alive
    torch.cuda.synchronize()
    env.close()


def test_cycle_length_schedule_does_not_change_results(monkeypatch):
    """The longest-cycle-first walk order (salp_robot.hip, robot_schedule_*) only moves envs between lanes:
    with it forced on and forced off every output and the final state are bit-identical."""
    n, seed, T = 5000, 9, 6
    rng = np.random.default_rng(4)
    acts = [np.stack([rng.uniform(-0.2, 1.2, n), rng.uniform(0, 0.3, n), rng.uniform(-1, 1, n)], axis=1).astype(np.float32)
            for _ in range(T)]
    acts[2][:7, 0] = np.nan          # a NaN action must not derail the schedule (bin 0)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SALP_ROBOT_SCHEDULE", flag)
        env = SalpRobotVectorEnv(n, device="cuda:0", seed=seed)
        rec = []
        for a in acts:
            obs, rew, term, trunc, info = env.step(a)
            rec.append([x.cpu().numpy().copy() for x in (obs, rew, term, trunc, info["inner_steps"])])
        rec.append([np.asarray(env.get_state())])
        outs.append(rec)
        env.close()
    for r0, r1 in zip(*outs):
        for x0, x1 in zip(r0, r1):
            assert np.array_equal(x0, x1, equal_nan=True)


def test_hip_robot_full_batch_sampled_parity():
    """65536 robots (above the size where the longest-cycle-first schedule switches on by itself), 8 cycles with
    cycle lengths from 0 to 14.5 s: 160 sampled robots against the oracle, each keyed by its global index."""
    n, seed, T = 65536, 21, 8
    env = SalpRobotVectorEnv(n, device="cuda:0", seed=seed)
    sample = np.unique(np.concatenate([np.arange(0, n, n // 80)[:80], np.arange(0, n, n // 80)[:80] + 1]))
    oracles = [rol.RobotOracleVec(1, seed=seed, env_index_base=int(i)) for i in sample]
    for o in oracles:
        o.reset(np.zeros(1, np.uint8))
    rng = np.random.default_rng(8)
    total_inner = 0
    for t in range(T):
        a = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1, n), rng.uniform(-1, 1, n)], axis=1).astype(np.float32)
        obs, rew, term, trunc, info = env.step(a)
        obs, rew, term, trunc = (x.cpu().numpy() for x in (obs, rew, term, trunc))
        inner = info["inner_steps"].cpu().numpy()
        total_inner += int(inner.sum())
        for j, i in enumerate(sample):
            ref = oracles[j].step(a[i:i + 1])
            assert inner[i] == ref["inner_steps"][0], (t, i)
            assert bool(term[i]) == bool(ref["terminated"][0]) and bool(trunc[i]) == bool(ref["truncated"][0]), (t, i)
            assert rel(obs[i], ref["obs"][0]) <= TOL and rel(rew[i], ref["reward"][0]) <= 1e-5, (t, i)
    state = env.get_state()
    for j, i in enumerate(sample):
        assert rel(state[:, i], oracles[j].get_state()[:, 0]) <= TOL, i
        oracles[j].close()
    assert total_inner > n * T * 300
    env.close()


def test_sac_learner_runs_on_the_robot_env():
    """train_robot.py's consumer side: the SAC learner (its hyper-parameters) on the batched HEAD simulator, eager and
    hipGraph-captured."""
    from underwater_swimmer_rl_amd.sac import SAC, SACConfig, train_sac, train_sac_graphed
    for run in (train_sac, train_sac_graphed):
        env = SalpRobotVectorEnv(256, device="cuda:0", seed=1)
        cfg = SACConfig.from_preset("salp_robot")
        cfg.learning_starts = 4
        agent = SAC(env.obs_dim, env.act_dim, cfg, device="cuda:0", seed=0,
                    act_low=env.single_action_space.low, act_high=env.single_action_space.high)
        m = run(env, agent, 24)
        assert m["vector_steps"] == 24 and m["updates"] == 20
        assert all(np.isfinite(m[k]) for k in ("critic_loss", "actor_loss", "alpha", "entropy"))
        env.close()
