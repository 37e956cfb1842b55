"""Multi-process coverage of the env-index sharding + all-gather path (world_size 2, gloo, CPU).

The per-rank simulator is injected: here it is the CPU oracle (test infrastructure), standing in
for the HIP engine so that the sharding / global-index keying / gather logic of
`ShardedSalpVectorEnv` runs without a GPU.  The product default engine is the HIP library."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd.sharded import ShardedSalpVectorEnv


class OracleEngine:
    """VectorEnv-shaped wrapper of the oracle with the HIP engine's call shapes."""

    def __init__(self, cfg, n, seed, base):
        self.o = ol.OracleVec(cfg, n, seed=seed, env_index_base=base)
        self.num_envs = n

    def reset(self, seed=None, options=None):
        return self.o.reset(), {}

    def step(self, actions):
        out = self.o.step(np.asarray(actions, np.float32), want_final=True)
        done = (out["terminated"] | out["truncated"]).astype(bool)
        fin = np.where(done[:, None], out["final_obs"], 0.0).astype(np.float32)   # rows of unfinished envs: unspecified
        info = {"food_collected": out["info"][:, 0], "steps_since_food": out["info"][:, 1], "collision": out["info"][:, 2],
                "final_observation": fin, "_final_observation": done}
        return out["obs"], out["reward"], out["terminated"].astype(bool), out["truncated"].astype(bool), info

    def rollout(self, actions=None, horizon=None):
        a = None if actions is None else np.asarray(actions, np.float32)
        return self.o.rollout(a, horizon=horizon)

    def close(self):
        self.o.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = pkg.load_env_config("sac_gail", max_steps_without_food=60)
        N, H, seed = 64, 150, 5
        rng = np.random.default_rng(0)
        act = rng.uniform(-1, 1, size=(H, N, 1)).astype(np.float32)          # the same GLOBAL actions on every rank
        env = ShardedSalpVectorEnv(cfg, N, seed=seed, engine_factory=lambda c, n, s, b: OracleEngine(c, n, s, b))
        assert env.local_envs == N // world and env.env_index_base == rank * (N // world)
        obs0, _ = env.reset()
        obs0 = obs0.clone()                 # gathered tensors are reused buffers (valid until the next call)
        steps = []
        for t in range(70):
            g_obs, g_rew, g_term, g_trunc, g_info = env.step(torch.from_numpy(act[t]))
            assert g_info["final_observation"].shape == (N, cfg.obs_dim) and g_info["food_collected"].shape == (N,)
            assert torch.equal(g_info["_final_observation"], g_term | g_trunc)
            steps.append((g_obs.numpy().copy(), g_rew.numpy().copy(), g_term.numpy().copy(), g_trunc.numpy().copy(),
                          g_info["final_observation"].numpy().copy(), g_info["food_collected"].numpy().copy(),
                          g_info["steps_since_food"].numpy().copy(), g_info["collision"].numpy().copy()))
        local, g_final = env.rollout(torch.from_numpy(act[70:]), gather="final")
        _, g_all = env.rollout(torch.from_numpy(act[:10]), gather="all", async_gather=True)
        env.wait_gather()
        _, g_none = env.rollout(torch.from_numpy(act[:5]), gather="none")
        assert g_none is None
        # three launches with the staged, double-buffered asynchronous gather in flight (bench.py's pattern):
        # slot 0, slot 1, slot 0 again; each gathered block must be its own launch's last observation
        asy = []
        for i in range(3):
            _, g = env.rollout(torch.from_numpy(act[20 + 4 * i: 24 + 4 * i]), gather="final", async_gather=True)
            if i == 1:
                env.wait_gather()
                asy.append(g.numpy().copy())      # launch 1, read before its slot partner is reused
            elif i == 2:
                env.wait_gather()
                asy.append(g.numpy().copy())      # launch 2 went through slot 0 again
        np.savez(os.path.join(tmp, f"rank{rank}.npz"), obs0=obs0.numpy(), g_final=g_final.numpy(), g_all=g_all.numpy(),
                 g_async1=asy[0], g_async2=asy[1],
                 **{f"s{t}_{k}": v for t, s in enumerate(steps)
                    for k, v in zip(("obs", "rew", "term", "trunc", "fin", "fc", "ssf", "coll"), s)})
        env.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for k in r0.files:   # every rank holds the same gathered batch
        assert np.array_equal(r0[k], r1[k], equal_nan=True), k
    # and it equals one unsharded simulator over all 64 envs
    cfg = pkg.load_env_config("sac_gail", max_steps_without_food=60)
    N, H, seed = 64, 150, 5
    act = np.random.default_rng(0).uniform(-1, 1, size=(H, N, 1)).astype(np.float32)
    one = ol.OracleVec(cfg, N, seed=seed)
    assert np.array_equal(r0["obs0"], one.reset())          # reset() draws new food, as snake:133-155 does
    finished = 0
    for t in range(70):
        out = one.step(act[t], want_final=True)
        assert np.array_equal(r0[f"s{t}_obs"], out["obs"]) and np.array_equal(r0[f"s{t}_rew"], out["reward"])
        assert np.array_equal(r0[f"s{t}_term"], out["terminated"].astype(bool))
        assert np.array_equal(r0[f"s{t}_trunc"], out["truncated"].astype(bool))
        # the gathered info is the unsharded engine's: counters of every env, terminal observation of finished ones
        assert np.array_equal(r0[f"s{t}_fc"], out["info"][:, 0]) and np.array_equal(r0[f"s{t}_ssf"], out["info"][:, 1])
        assert np.array_equal(r0[f"s{t}_coll"], out["info"][:, 2])
        done = (out["terminated"] | out["truncated"]).astype(bool)
        finished += int(done.sum())
        assert np.array_equal(r0[f"s{t}_fin"][done], out["final_obs"][done])
    assert finished > 0                                                        # envs of both halves finished (truncation at step 61) in the stepped part
    out = one.rollout(act[70:])
    assert out["truncated"].sum() > 0                                         # resets happened inside the shard
    assert np.array_equal(r0["g_final"], out["obs"][-1])
    out = one.rollout(act[:10])
    g_all = r0["g_all"]                                                      # [G, H, n_local, D]
    assert g_all.shape == (2, 10, 32, cfg.obs_dim)
    assert np.array_equal(np.concatenate([g_all[0], g_all[1]], axis=1), out["obs"])
    one.rollout(act[:5])                                                      # the gather="none" launch
    one.rollout(act[20:24])
    assert np.array_equal(r0["g_async1"], one.rollout(act[24:28])["obs"][-1])
    assert np.array_equal(r0["g_async2"], one.rollout(act[28:32])["obs"][-1])


def _sac_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from underwater_swimmer_rl_amd.sac import SAC, SACConfig
        torch.manual_seed(100 + rank)                       # replicas start from DIFFERENT weights ...
        cfg = SACConfig(hidden_sizes=(32, 32), batch_size=64, alpha=None)
        agent = SAC(24, 1, cfg, device="cpu", data_parallel=True)
        flat = lambda: torch.cat([p.detach().reshape(-1) for p in [*agent.actor.parameters(), *agent.critic.parameters(),
                                                                    *agent.critic_target.parameters(), agent.log_alpha]])
        start = flat().clone()
        g = torch.Generator().manual_seed(7 + rank)         # ... and learn from DIFFERENT data
        for _ in range(6):
            batch = (torch.randn(64, 24, generator=g), torch.rand(64, 1, generator=g) * 2 - 1, torch.randn(64, generator=g),
                     torch.randn(64, 24, generator=g), (torch.rand(64, generator=g) > 0.9).float())
            agent.update(batch)
        torch.save({"start": start, "end": flat(), "updates": agent.updates}, os.path.join(tmp, f"sac{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_sac_replicas_stay_identical(tmp_path):
    """SAC(data_parallel=True): broadcast at construction, one gradient all-reduce (mean) per backward."""
    port = _free_port()
    mp.spawn(_sac_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "sac0.pt"), weights_only=True)
    b = torch.load(os.path.join(tmp_path, "sac1.pt"), weights_only=True)
    assert torch.equal(a["start"], b["start"])                 # rank 0's weights everywhere
    assert not torch.equal(a["start"], a["end"])               # learning happened
    assert torch.equal(a["end"], b["end"])                     # and the replicas never diverged
    assert a["updates"] == b["updates"] == 6
