"""hipGraph capture of the acting loop (policy -> salp_vec_step) through the C ABI: replays must produce
exactly what the same calls issued one by one produce."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _policy(torch, obs_dim, act_dim, device):
    g = torch.Generator().manual_seed(5)
    w1 = (torch.randn(obs_dim, 32, generator=g) * 0.5).to(device)
    w2 = (torch.randn(32, act_dim, generator=g) * 0.5).to(device)
    return lambda o: torch.tanh(torch.tanh(o @ w1) @ w2)


@pytest.mark.parametrize("preset,n", [("single_food", 1000), ("sac_gail", 257)])
def test_graph_replay_matches_eager(preset, n):
    import torch
    from underwater_swimmer_rl_amd import SalpVectorEnv
    K, R = 8, 6
    eager = SalpVectorEnv(preset, num_envs=n, seed=11)
    graphed = SalpVectorEnv(preset, num_envs=n, seed=11)
    pol = _policy(torch, eager.obs_dim, eager.act_dim, eager.device)
    o_e, _ = eager.reset()
    o_g, _ = graphed.reset()
    assert torch.equal(o_e, o_g)

    rec = dict(obs=torch.zeros(K, n, graphed.obs_dim, device=graphed.device),
               rew=torch.zeros(K, n, device=graphed.device),
               done=torch.zeros(K, n, dtype=torch.bool, device=graphed.device))

    def record(k, obs, act, rew, term, trunc, info):
        rec["obs"][k].copy_(obs)
        rec["rew"][k].copy_(rew)
        torch.logical_or(term, trunc, out=rec["done"][k])

    g = graphed.capture_policy_steps(pol, n_steps=K, record=record)
    # capture (and its warm-up) must leave the envs where they were
    f_e, i_e = eager.get_state()
    f_g, i_g = graphed.get_state()
    assert np.array_equal(f_e, f_g, equal_nan=True) and np.array_equal(i_e, i_g)

    for r in range(R):
        g.replay()
        torch.cuda.synchronize()
        for k in range(K):
            seen = o_e.clone()
            o_e, rew, term, trunc, info = eager.step(pol(seen))
            assert torch.equal(rec["obs"][k], seen), (r, k)
            assert torch.equal(rec["rew"][k], rew), (r, k)
            assert torch.equal(rec["done"][k], term | trunc), (r, k)
        o_last = graphed._step_cache["obs"]
        assert torch.equal(o_last, o_e)
        d = info["_final_observation"]          # final_observation rows are defined for finished envs only
        assert torch.equal(graphed._step_cache["info"]["_final_observation"], d)
        assert torch.equal(graphed._step_cache["info"]["final_observation"][d], info["final_observation"][d])
    f_e, i_e = eager.get_state()
    f_g, i_g = graphed.get_state()
    assert np.array_equal(f_e, f_g, equal_nan=True) and np.array_equal(i_e, i_g)
    eager.close(); graphed.close()
