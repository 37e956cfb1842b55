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


@pytest.mark.parametrize("preset,n", [("single_food", 1000), ("sac_gail", 257)])
def test_reseed_keeps_captured_graphs_valid(preset, n):
    """reset(seed=s) re-keys the draw streams IN PLACE (salp_vec_reseed: no free / realloc, no launch parameter changes —
    the kernels read the key words from device memory), so a hipGraph captured before it replays correctly after it:
    capture with seed 11 -> reset(seed=29) -> replays == an eager env created with seed 29.  Also: the re-seeded handle is
    exactly a freshly created one (state, counters, statistics), and reset(seed=s) twice gives identical episodes
    (reference: snake:133-155 reset(seed), legacy:95-96)."""
    import torch
    from underwater_swimmer_rl_amd import SalpVectorEnv
    K, R = 8, 4
    graphed = SalpVectorEnv(preset, num_envs=n, seed=11, max_steps_without_food=20)   # frequent truncations: resets draw
    pol = _policy(torch, graphed.obs_dim, graphed.act_dim, graphed.device)
    graphed.reset()
    rec = dict(obs=torch.zeros(K, n, graphed.obs_dim, device=graphed.device), rew=torch.zeros(K, n, device=graphed.device))

    def record(k, obs, act, rew, term, trunc, info):
        rec["obs"][k].copy_(obs)
        rec["rew"][k].copy_(rew)

    g = graphed.capture_policy_steps(pol, n_steps=K, record=record)
    for _ in range(3):                      # advance under the OLD seed first
        g.replay()
    h_before = graphed._lib._h.value
    ptr_before = graphed._step_cache["obs"].data_ptr()
    o_g, _ = graphed.reset(seed=29)
    assert graphed._lib._h.value == h_before and graphed._step_cache["obs"].data_ptr() == ptr_before
    assert graphed._lib.global_step == 0 and graphed._lib.stats()["env_steps"] == 0

    eager = SalpVectorEnv(preset, num_envs=n, seed=29, max_steps_without_food=20)
    o_e, _ = eager.reset()                  # a fresh handle is already reset; reset() again continues its streams ...
    o_g2, _ = graphed.reset()               # ... so the re-seeded one does the same
    assert torch.equal(o_e, o_g2)
    f_e, i_e = eager.get_state(); f_g, i_g = graphed.get_state()
    assert np.array_equal(f_e, f_g, equal_nan=True) and np.array_equal(i_e, i_g)
    finished = 0
    for r in range(R):
        g.replay()
        torch.cuda.synchronize()
        for k in range(K):
            seen = o_e.clone()
            o_e, rew, term, trunc, info = eager.step(pol(seen))
            finished += int((term | trunc).sum())
            assert torch.equal(rec["obs"][k], seen), (r, k)
            assert torch.equal(rec["rew"][k], rew), (r, k)
    assert finished > 0                      # autoresets (placement draws under the new key) were exercised
    f_e, i_e = eager.get_state(); f_g, i_g = graphed.get_state()
    assert np.array_equal(f_e, f_g, equal_nan=True) and np.array_equal(i_e, i_g)
    assert graphed._lib.stats() == eager._lib.stats()

    # reset(seed=s) twice: identical episodes
    a0, _ = graphed.reset(seed=5); a0 = a0.clone()
    traj = []
    for _ in range(20):
        o, rwd, *_ = graphed.step(pol(graphed._step_cache["obs"] if traj else a0))
        traj.append((o.clone(), rwd.clone()))
    b0, _ = graphed.reset(seed=5)
    assert torch.equal(a0, b0)
    for t in range(20):
        o, rwd, *_ = graphed.step(pol(graphed._step_cache["obs"] if t else b0))
        assert torch.equal(o, traj[t][0]) and torch.equal(rwd, traj[t][1])
    # a seed cannot be combined with a partial mask
    m = torch.zeros(n, dtype=torch.uint8); m[0] = 1
    with pytest.raises(ValueError):
        graphed.reset(seed=1, mask=m)
    eager.close(); graphed.close()
