"""BASELINE.json configs[2] at full size AND full horizon: 262144 envs x 5000 steps
(single_food_long_horizon) in 20 fused launches on the device, checked against the oracle on a strided
sample of envs for every step, plus whole-batch accounting.  Outputs stay on the GPU (a 5000-step block
would be 126 TB); only the sampled columns are copied back."""
import numpy as np
import pytest

import oracle_lib as ol
import underwater_swimmer_rl_amd as pkg
from golden_util import obs_diff

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_config3_full_size_full_horizon_sampled_parity():
    cfg = pkg.load_env_config("single_food_long_horizon")
    n, H, launches, seed = 262144, 250, 20, 0
    env = pkg.SalpVectorEnv(cfg, n, device="cuda:0", seed=seed)
    sample = np.arange(0, n, n // 96)[:96]                      # 96 envs spread over the batch
    sidx = torch.as_tensor(sample, device="cuda")
    oracles = [ol.OracleVec(cfg, 1, seed=seed, env_index_base=int(i)) for i in sample]
    g = torch.Generator(device="cuda").manual_seed(7)
    worst_obs, worst_rew, dones = 0.0, 0.0, 0
    for k in range(launches):
        act = torch.rand((H, n, 1), generator=g, device="cuda") * 2 - 1
        out = env.rollout(act)
        a_s = act[:, sidx].cpu().numpy()
        o_s = out["obs"][:, sidx].cpu().numpy()
        r_s = out["reward"][:, sidx].cpu().numpy().astype(np.float64)
        t_s = out["terminated"][:, sidx].cpu().numpy()
        u_s = out["truncated"][:, sidx].cpu().numpy()
        for j, orc in enumerate(oracles):
            ref = orc.rollout(np.ascontiguousarray(a_s[:, j:j + 1]))
            assert np.array_equal(t_s[:, j], ref["terminated"][:, 0]), (k, j)
            assert np.array_equal(u_s[:, j], ref["truncated"][:, 0]), (k, j)
            worst_obs = max(worst_obs, float(obs_diff(cfg, o_s[:, j], ref["obs"][:, 0]).max()))
            rr = ref["reward64"][:, 0]
            worst_rew = max(worst_rew, float((np.abs(r_s[:, j] - rr) / np.maximum(1.0, np.abs(rr))).max()))
            dones += int(ref["terminated"].sum() + ref["truncated"].sum())
        # size-independent properties of the whole batch, on the device
        o = out["obs"]
        assert bool(torch.isfinite(o).all())
        assert float(o[..., 9].abs().max()) <= 1.0 + 1e-6 and float(o[..., 8].min()) >= 0.0
        assert float(o[..., 0].min()) > 0.09 and float(o[..., 0].max()) < 0.91
    assert worst_obs <= 1e-5 and worst_rew <= 1e-5, (worst_obs, worst_rew)
    assert dones > 200                                            # the sample itself saw hundreds of resets
    st = env.stats()
    assert st["env_steps"] == n * H * launches
    assert st["episodes"] == st["terminated"] + st["truncated"] and st["collisions"] == st["terminated"]
    print(f"5000 steps x {n} envs: sampled max obs diff {worst_obs:.3g}, reward {worst_rew:.3g}, "
          f"sample resets {dones}, batch episodes {st['episodes']}, food {st['food_collected']}")
    env.close()


def test_multi_food_full_size_sampled_parity():
    """sac_gail (12 foods: LDS-resident foods, wavefront-cooperative respawn / reset placement) at 262144 envs x
    2000 steps: a strided sample of whole wavefronts' neighbours against the oracle on every step — the sampled
    envs respawn and reset while other lanes of their wavefronts do too."""
    cfg = pkg.load_env_config("sac_gail")
    n, H, launches, seed = 262144, 250, 8, 3
    env = pkg.SalpVectorEnv(cfg, n, device="cuda:0", seed=seed)
    base = np.arange(0, n, n // 24)[:24]
    sample = np.unique(np.concatenate([base, base + 1, base + 63]))   # neighbours inside one wavefront too
    sample = sample[sample < n]
    sidx = torch.as_tensor(sample, device="cuda")
    oracles = [ol.OracleVec(cfg, 1, seed=seed, env_index_base=int(i)) for i in sample]
    g = torch.Generator(device="cuda").manual_seed(11)
    worst_obs, worst_rew, dones, foods = 0.0, 0.0, 0, 0
    for k in range(launches):
        act = torch.rand((H, n, 1), generator=g, device="cuda") * 2 - 1
        out = env.rollout(act)
        a_s = act[:, sidx].cpu().numpy()
        o_s = out["obs"][:, sidx].cpu().numpy()
        r_s = out["reward"][:, sidx].cpu().numpy().astype(np.float64)
        t_s = out["terminated"][:, sidx].cpu().numpy()
        u_s = out["truncated"][:, sidx].cpu().numpy()
        for j, orc in enumerate(oracles):
            ref = orc.rollout(np.ascontiguousarray(a_s[:, j:j + 1]))
            assert np.array_equal(t_s[:, j], ref["terminated"][:, 0]), (k, j)
            assert np.array_equal(u_s[:, j], ref["truncated"][:, 0]), (k, j)
            worst_obs = max(worst_obs, float(obs_diff(cfg, o_s[:, j], ref["obs"][:, 0]).max()))
            rr = ref["reward64"][:, 0]
            worst_rew = max(worst_rew, float((np.abs(r_s[:, j] - rr) / np.maximum(1.0, np.abs(rr))).max()))
            dones += int(ref["terminated"].sum() + ref["truncated"].sum())
            foods += int((rr > 5.0).sum())
        o = out["obs"]
        assert bool(torch.isfinite(o).all())
        assert float(o[..., 22].max()) <= 1.0 and float(o[..., 22].min()) >= 0.0
    assert worst_obs <= 1e-5 and worst_rew <= 1e-5, (worst_obs, worst_rew)
    assert dones > 50 and foods > 50
    # the device state of the sampled envs (fp64, incl. all 12 food positions) still matches the oracle
    f64, i32 = env.get_state()
    for j, orc in enumerate(oracles):
        of, oi = orc.get_state()
        assert np.array_equal(i32[:, sample[j]], oi[:, 0]), j
        a, b = f64[:, sample[j]], of[:, 0]
        assert np.array_equal(np.isnan(a), np.isnan(b)), j
        assert float(np.nanmax(np.abs(a - b))) <= 1e-6, j
    st = env.stats()
    assert st["env_steps"] == n * H * launches and st["episodes"] == st["terminated"] + st["truncated"]
    print(f"sac_gail 2000 steps x {n} envs: sampled max obs diff {worst_obs:.3g}, reward {worst_rew:.3g}, "
          f"sample resets {dones}, captures {foods}, batch episodes {st['episodes']}, food {st['food_collected']}")
    env.close()
