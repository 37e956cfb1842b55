"""BASELINE.json configs[3]'s PER-GPU workload on the HIP path: shard 7 of the 8-way split of 1 048 576 envs —
131072 envs at env_index_base = 7 x 131072, single_food_long_horizon, fused 250-step launches — through
ShardedSalpVectorEnv on RCCL (world size 1 on this one-GPU box, `rehearse_shard=(7, 8)`), with the exchange pattern of
`bench.py --gpus 8 --gather all` (every launch all-gathers every observation it returned, double-buffered, in pieces of
at most ~2 GB).  Checked against the oracle on a strided sample of envs under their GLOBAL indices, plus size-independent
properties of the whole shard.  (The 8-GPU job itself is the driver's; reference semantics: snake:133-155 per env,
independent of the sharding.)"""
import os

import numpy as np
import pytest

import oracle_lib as ol
import underwater_swimmer_rl_amd as pkg
from golden_util import obs_diff

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_configs3_shard7_of_8_workload_with_gather_all():
    import torch.distributed as dist
    from underwater_swimmer_rl_amd.sharded import ShardedSalpVectorEnv
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29537")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = pkg.load_env_config("single_food_long_horizon")
        total, shards, shard, H, launches, seed = 1048576, 8, 7, 250, 2, 0
        senv = ShardedSalpVectorEnv(cfg, total, device="cuda:0", seed=seed, rehearse_shard=(shard, shards))
        n = senv.local_envs
        assert n == 131072 and senv.env_index_base == shard * 131072
        env = senv.engine
        dev = env.device
        # the bench's buffers: two output blocks, the collective of launch k reading block k % 2 beside launch k + 1
        outs = [dict(obs=torch.empty((H, n, cfg.obs_dim), device=dev), reward=torch.empty((H, n), device=dev),
                     terminated=torch.empty((H, n), dtype=torch.uint8, device=dev),
                     truncated=torch.empty((H, n), dtype=torch.uint8, device=dev)) for _ in range(2)]
        gather_steps = max(1, min(H, int((2 << 30) // (n * cfg.obs_dim * 4))))
        works = [None, None]
        sample = np.arange(5, n, n // 48)[:48]                       # 48 envs spread over the shard
        sidx = torch.as_tensor(sample, device=dev)
        oracles = [ol.OracleVec(cfg, 1, seed=seed, env_index_base=senv.env_index_base + int(i)) for i in sample]
        g = torch.Generator(device=dev).manual_seed(1234 + shard)
        worst_obs = worst_rew = 0.0
        gathered_ok = 0
        for k in range(launches):
            b = k & 1
            if works[b] is not None:
                for w in works[b]:
                    w.wait()
            act = torch.rand((H, n, 1), generator=g, device=dev) * 2 - 1
            out = env.rollout(act, out=outs[b])
            works[b] = []
            pieces = []
            for i, h0 in enumerate(range(0, H, gather_steps)):
                piece = out["obs"][h0:h0 + gather_steps]
                gt, w = senv.all_gather(f"all_obs{b}_{i}", piece.reshape(1, piece.shape[0], n, cfg.obs_dim), async_op=True)
                works[b].append(w)
                pieces.append((h0, gt))
            for w in works[b]:
                w.wait()
            works[b] = None
            for h0, gt in pieces:                                     # world size 1: the gathered block is the local one
                assert gt.shape[0] == 1 and torch.equal(gt[0], out["obs"][h0:h0 + gt.shape[1]])
                gathered_ok += 1
            ll = env._lib.last_launch()
            assert ll["food_slots"] == 1 and ll["literal_constants"] == 1 and ll["full_signature"] == 1
            assert ll["envs_unpredicated"] == n and ll["envs_predicated"] == 0
            a_s = act[:, sidx].cpu().numpy()
            o_s = out["obs"][:, sidx].cpu().numpy()
            r_s = out["reward"][:, sidx].cpu().numpy().astype(np.float64)
            t_s = out["terminated"][:, sidx].cpu().numpy()
            u_s = out["truncated"][:, sidx].cpu().numpy()
            for j, orc in enumerate(oracles):
                ref = orc.rollout(np.ascontiguousarray(a_s[:, j:j + 1]))
                assert np.array_equal(t_s[:, j], ref["terminated"][:, 0]) and np.array_equal(u_s[:, j], ref["truncated"][:, 0]), (k, j)
                worst_obs = max(worst_obs, float(obs_diff(cfg, o_s[:, j], ref["obs"][:, 0]).max()))
                rr = ref["reward64"][:, 0]
                worst_rew = max(worst_rew, float((np.abs(r_s[:, j] - rr) / np.maximum(1.0, np.abs(rr))).max()))
            o = out["obs"]                                            # size-independent properties of the whole shard
            assert bool(torch.isfinite(o).all())
            assert float(o[..., 9].abs().max()) <= 1.0 + 1e-6 and float(o[..., 8].min()) >= 0.0
            assert float(o[..., 0].min()) > 0.09 and float(o[..., 0].max()) < 0.91
        assert worst_obs <= 1e-5 and worst_rew <= 1e-5, (worst_obs, worst_rew)
        assert gathered_ok >= launches
        st = env.stats()
        assert st["env_steps"] == n * H * launches and st["episodes"] == st["terminated"] + st["truncated"]
        # the shard's envs are NOT the first 131072 envs of the job: env 5 of this shard differs from global env 5
        first = pkg.SalpVectorEnv(cfg, 64, device="cuda:0", seed=seed, env_index_base=0)
        o_here, _ = senv.engine.reset()
        o_first, _ = first.reset()
        assert not torch.equal(o_here[:64, 10:12], o_first[:, 10:12])
        first.close()
        senv.close()
    finally:
        dist.destroy_process_group()
