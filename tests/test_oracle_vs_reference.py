"""Live cross-check of the CPU oracle against the reference's Python implementation (only where
/root/reference exists, i.e. the build container): longer randomized runs than the committed vectors."""
import numpy as np
import pytest

import oracle_lib as ol
import ref_harness as rh
import underwater_swimmer_rl_amd as pkg

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present on this machine")


@pytest.mark.parametrize("preset,over", [
    ("single_food", {}), ("single_food_long_horizon", {}), ("sac_gail", {}),
    ("single_food", dict(forced_breathing=False)),
    ("sac_gail", dict(num_food_items=4, random_food_count=True, max_steps_without_food=80)),
])
def test_bit_exact_against_reference(preset, over):
    cfg = pkg.load_env_config(preset, **over)
    n, T, seed = 3, 2500, 4321
    orc = ol.OracleVec(cfg, n, seed=seed)
    refs = [rh.ReferenceEnv(seed, i, **cfg.env_kwargs()) for i in range(n)]
    assert np.array_equal(orc.observe(), np.stack([r.reset() for r in refs]))
    rng = np.random.default_rng(11)
    act = rng.uniform(-1, 1, size=(T, n, cfg.act_dim)).astype(np.float32)
    if not cfg.forced_breathing:
        act[..., 0] = rng.uniform(0, 1, size=(T // 25 + 1, n)).repeat(25, axis=0)[:T]
    out = orc.rollout(act, want_final=True)
    for t in range(T):
        for i, r in enumerate(refs):
            o, rew, term, trunc, info = r.step(act[t, i])
            assert rew == out["reward64"][t, i], (t, i)
            assert bool(term) == bool(out["terminated"][t, i]) and bool(trunc) == bool(out["truncated"][t, i]), (t, i)
            if term or trunc:
                assert np.array_equal(o, out["final_obs"][t, i])
                o = r.reset()
            assert np.array_equal(o, out["obs"][t, i]), (t, i)


def test_presets_match_reference_yaml():
    """config.PRESETS restates configs/*.yaml `environment.params`."""
    import os
    for name in ("single_food", "single_food_long_horizon", "sac_gail", "defaults"):
        a = pkg.load_env_config(name)
        b = pkg.load_env_config(os.path.join(rh.REFERENCE_ROOT, "configs", f"{name}.yaml"))
        assert a == b, name
