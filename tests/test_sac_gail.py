"""The learner side: SAC with SB3's MlpPolicy architecture and the GAIL
discriminator, on device tensors.  CPU tests check shapes / maths on tiny problems; the GPU test
runs the learner against the HIP vector env."""
import math

import numpy as np
import pytest
import torch

import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd.gail import Discriminator, ExpertBuffer, gail_reward_fn
from underwater_swimmer_rl_amd.sac import SAC, Actor, DeviceReplayBuffer, SACConfig, TwinQ, train_sac


def test_actor_matches_sb3_mlp_policy_shapes_and_squashing():
    torch.manual_seed(0)
    a = Actor(24, 2, (256, 256), act_low=[0.0, -1.0], act_high=[1.0, 1.0])
    assert sum(p.numel() for p in a.parameters()) == 24 * 256 + 256 + 256 * 256 + 256 + 2 * (256 * 2 + 2)
    act, logp = a(torch.randn(512, 24))
    assert act.shape == (512, 2) and logp.shape == (512,)
    assert act[:, 0].min() >= 0.0 and act[:, 0].max() <= 1.0 and act[:, 1].abs().max() <= 1.0
    # log-prob of the squashed Gaussian agrees with the direct formula
    obs = torch.randn(64, 24)
    torch.manual_seed(1)
    act, logp = a(obs)
    h = a.body(obs)
    mu, ls = a.mu(h), a.log_std(h).clamp(-20, 2)
    u = torch.atanh(a.unscale(act).clamp(-1 + 1e-6, 1 - 1e-6))
    direct = torch.distributions.Normal(mu, ls.exp()).log_prob(u).sum(-1) - torch.log(1 - torch.tanh(u) ** 2 + 1e-6).sum(-1)
    assert torch.allclose(logp, direct, atol=2e-3)
    q = TwinQ(24, 2)
    q1, q2 = q(obs, act)
    assert q1.shape == (64,) and q2.shape == (64,)


def test_replay_buffer_ring_semantics():
    buf = DeviceReplayBuffer(10, 3, 1, "cpu")
    for k in range(4):
        n = 4
        o = torch.full((n, 3), float(k))
        buf.add(o, torch.zeros(n, 1), torch.full((n,), float(k)), o + 1, torch.zeros(n, dtype=torch.bool))
    assert buf.size == 10 and buf.pos == 6
    assert set(buf.rew.tolist()) == {1.0, 2.0, 3.0}      # the oldest rows (k = 0) were overwritten
    o, a, r, no, t = buf.sample(32)
    assert o.shape == (32, 3) and torch.equal(no, o + 1)


def test_sac_learns_a_one_step_bandit():
    """Reward = -(a - 0.5)^2: the squashed policy mean must move to 0.5."""
    torch.manual_seed(0)
    cfg = SACConfig(hidden_sizes=(32, 32), batch_size=256, alpha=0.01, learning_rate=3e-3)
    agent = SAC(4, 1, cfg, device="cpu")
    buf = DeviceReplayBuffer(4096, 4, 1, "cpu")
    obs = torch.zeros(4096, 4)
    act = torch.rand(4096, 1) * 2 - 1
    buf.add(obs, act, -(act[:, 0] - 0.5) ** 2, obs, torch.ones(4096, dtype=torch.bool))
    for _ in range(400):
        m = agent.update(buf.sample(cfg.batch_size))
    a = agent.act(torch.zeros(1, 4), deterministic=True)
    assert abs(float(a) - 0.5) < 0.15, float(a)
    assert math.isfinite(float(m["critic_loss"]))
    sd = agent.state_dict()
    other = SAC(4, 1, cfg, device="cpu")
    other.load_state_dict(sd)
    assert torch.equal(other.act(torch.zeros(1, 4), deterministic=True), a)


def test_discriminator_separates_expert_from_agent_and_rewards_follow():
    torch.manual_seed(0)
    d = Discriminator(6, 1, (32, 32), learning_rate=3e-3, device="cpu")
    eb = ExpertBuffer(6, 1, device="cpu")
    eo = np.random.default_rng(0).normal(1.0, 0.3, size=(500, 6)).astype(np.float32)
    eb.add_episode(eo, np.full((500, 1), 0.8, np.float32))
    assert len(eb) == 500 and eb.episodes == 1
    for _ in range(300):
        agent_batch = {"observations": torch.randn(64, 6) * 0.3 - 1.0, "actions": torch.full((64, 1), -0.8)}
        m = d.update(eb.sample(64), agent_batch)
    assert m["discriminator_accuracy"] > 0.95
    r_exp = d.predict_reward(torch.full((8, 6), 1.0), torch.full((8, 1), 0.8))
    r_agt = d.predict_reward(torch.full((8, 6), -1.0), torch.full((8, 1), -0.8))
    assert r_exp.shape == (8, 1) and float(r_exp.mean()) > float(r_agt.mean()) + 1.0      # [B, 1] as the reference
    mix = gail_reward_fn(d)(torch.full((8, 6), 1.0), torch.full((8, 1), 0.8), torch.ones(8))
    assert mix.shape == (8,) and torch.allclose(mix, 0.3 * torch.ones(8) + 0.7 * r_exp.squeeze(-1))
    host = Discriminator.metrics_to_host(m)
    assert set(host) == {"discriminator_loss", "expert_loss", "agent_loss", "discriminator_accuracy",
                         "expert_prob_mean", "agent_prob_mean"} and all(isinstance(v, float) for v in host.values())


def test_agent_presets_restate_the_yaml_agent_blocks():
    c = SACConfig.from_preset("single_food_long_horizon")
    assert (c.batch_size, c.buffer_size, c.gamma, c.alpha, c.target_entropy) == (256, 100_000, 0.995, 0.2, -0.5)
    assert SACConfig.from_preset("sac_gail").alpha == 0.1


@pytest.mark.gpu
def test_sac_trains_on_the_hip_vector_env():
    """configs[4] plumbing: SAC (sac_gail.yaml agent block) on the HIP VectorEnv, GAIL reward mixed in."""
    env = pkg.SalpVectorEnv("sac_gail", num_envs=1024, device="cuda:0", seed=0)
    cfg = SACConfig.from_preset("sac_gail")
    cfg.learning_starts, cfg.updates_per_step = 20, 2
    agent = SAC(env.obs_dim, env.act_dim, cfg, device="cuda:0", seed=0)
    disc = Discriminator(env.obs_dim, env.act_dim, device="cuda:0")
    eb = ExpertBuffer(env.obs_dim, env.act_dim, device="cuda:0")
    eb.add_episode(np.random.default_rng(0).uniform(-1, 1, (256, env.obs_dim)), np.zeros((256, env.act_dim)))
    m = train_sac(env, agent, total_vector_steps=60, reward_fn=gail_reward_fn(disc))
    assert m["env_steps"] == 60 * 1024 and m["updates"] == 80
    assert all(math.isfinite(m[k]) for k in ("critic_loss", "actor_loss", "alpha"))
    d = disc.update(eb.sample(128), {"observations": env.observe().clone(), "actions": agent.act(env.observe())})
    assert math.isfinite(float(d["discriminator_loss"]))
    env.close()


def test_discriminator_on_the_reference_human_demos():
    """GAIL on the reference's own recorded demonstrations (tests/golden/human_demo_*.npz: first 1500
    (obs[24], action) pairs of each of data/expert_demos/human/*.pkl)."""
    import os
    torch.manual_seed(0)
    d = Discriminator(24, 1, (64, 64), learning_rate=1e-3, device="cpu")
    eb = ExpertBuffer(24, 1, device="cpu")
    n = eb.load_directory(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"), "human_demo_*.npz")
    assert n == 5 and len(eb) == 5 * 1500
    b = eb.sample(256)
    assert b["observations"].shape == (256, 24) and b["actions"].abs().max() <= 1.0
    assert b["observations"][:, 6].min() >= 1.0            # body size column of real observations
    for _ in range(200):       # "agent" = uniformly random observations / actions
        m = d.update(eb.sample(128), {"observations": torch.rand(128, 24) * 2 - 1, "actions": torch.rand(128, 1) * 2 - 1})
    assert m["discriminator_accuracy"] > 0.9


def test_capturable_replay_ops_match_the_eager_ring():
    """add_capturable / sample_capturable (device-side cursor) against add (Python cursor)."""
    from underwater_swimmer_rl_amd.sac import DeviceReplayBuffer
    g = torch.Generator().manual_seed(0)
    a, b = DeviceReplayBuffer(10, 3, 1, "cpu"), DeviceReplayBuffer(10, 3, 1, "cpu")
    for _ in range(7):      # wraps several times
        rows = [torch.randn(4, 3, generator=g), torch.randn(4, 1, generator=g), torch.randn(4, generator=g),
                torch.randn(4, 3, generator=g), torch.rand(4, generator=g) > 0.5]
        a.add(*rows)
        b.add_capturable(*rows); b.advance_host(4)
        assert (a.pos, a.size) == (b.pos, b.size) == (int(b.pos_t), int(b.size_t))
        for x, y in ((a.obs, b.obs), (a.act, b.act), (a.rew, b.rew), (a.next_obs, b.next_obs), (a.term, b.term)):
            assert torch.equal(x[:a.size], y[:b.size])
    o, ac, r, no, te = b.sample_capturable(64)
    assert o.shape == (64, 3) and te.shape == (64,)


@pytest.mark.gpu
def test_graphed_sac_training_runs_and_counts_like_the_eager_loop():
    """train_sac_graphed: one hipGraph replay per vector step; same bookkeeping as train_sac."""
    import underwater_swimmer_rl_amd as salp
    from underwater_swimmer_rl_amd.sac import SAC, SACConfig, train_sac_graphed
    env = salp.SalpVectorEnv("sac_gail", num_envs=512, device="cuda:0", seed=3)
    cfg = SACConfig.from_preset("sac_gail")
    cfg.learning_starts = 6
    agent = SAC(env.obs_dim, env.act_dim, cfg, device="cuda:0", seed=0,
                act_low=env.single_action_space.low, act_high=env.single_action_space.high)
    w0 = [p.detach().clone() for p in agent.actor.parameters()]
    m = train_sac_graphed(env, agent, 40, poll_every=4)
    assert m["vector_steps"] == 40 and m["env_steps"] == 40 * 512
    assert m["updates"] == 34 and m["graphs"] == 2          # steps 6..39 learn; one graph per phase
    assert env.stats()["env_steps"] == 40 * 512             # every replay really stepped the envs
    assert all(np.isfinite(m[k]) for k in ("critic_loss", "actor_loss", "alpha", "entropy"))
    assert any(not torch.equal(a, b) for a, b in zip(w0, agent.actor.parameters()))
    env.close()


@pytest.mark.gpu
def test_configs4_first_food_capture_with_the_yaml_agent_block():
    """BASELINE configs[4] on one GPU: SAC with configs/sac_gail.yaml's agent block (SACConfig.from_preset) on 4096
    envs of the sac_gail preset, one hipGraph replay per vector step, fixed seeds: some env captures its first food
    within a few hundred vector steps (164 when DESIGN.md §8 was written); wall-clock is reported, not asserted."""
    import json
    from underwater_swimmer_rl_amd.sac import train_sac_graphed
    env = pkg.SalpVectorEnv("sac_gail", num_envs=4096, device="cuda:0", seed=0)
    cfg = SACConfig.from_preset("sac_gail")
    cfg.learning_starts = 50
    agent = SAC(env.obs_dim, env.act_dim, cfg, device="cuda:0", seed=0,
                act_low=env.single_action_space.low, act_high=env.single_action_space.high)
    m = train_sac_graphed(env, agent, 600, stop_at_first_food=True)
    print("configs[4] first capture:", json.dumps({k: m[k] for k in ("first_food_vector_step", "first_food_wall_s", "wall_s", "vector_steps")}))
    assert m["first_food_vector_step"] is not None and 1 <= m["first_food_vector_step"] <= 400
    assert m["first_food_wall_s"] is not None and m["first_food_wall_s"] < 60.0
    assert env.stats()["food_collected"] >= 1
    env.close()


@pytest.mark.gpu
def test_segmented_graph_iteration_trains_like_the_single_graph():
    """The data-parallel form of the captured loop (graph segments with the gradient all-reduces between them), forced
    at world size 1: same number of updates, finite losses, and — the all-reduce being the identity at world size 1 —
    the same first-capture step as the single-graph loop from the same seeds."""
    from underwater_swimmer_rl_amd.sac import train_sac_graphed
    out = []
    for seg in (False, True):
        env = pkg.SalpVectorEnv("sac_gail", num_envs=1024, device="cuda:0", seed=0)
        cfg = SACConfig.from_preset("sac_gail")
        cfg.learning_starts, cfg.updates_per_step = 20, 2
        agent = SAC(env.obs_dim, env.act_dim, cfg, device="cuda:0", seed=0)
        m = train_sac_graphed(env, agent, 80, force_segments=seg)
        assert m["segmented"] is seg and m["updates"] == 2 * 60
        assert all(math.isfinite(m[k]) for k in ("critic_loss", "actor_loss", "alpha", "entropy"))
        assert m["graphs"] == (2 if not seg else 1 + (1 + 2 * 2))       # random phase: 1; learning phase: 1 or 1 + 2 x updates
        out.append(m)
        env.close()
    assert out[0]["env_steps"] == out[1]["env_steps"] == 80 * 1024
