"""Multi-GPU sharding of the batched simulator: one process per GPU, envs split by index.

The reference is single-process (no distributed code); this is the north_star's
"sharded across the 8 GPUs of one node by env-index with an RCCL all-gather over xGMI of
returned observations".  Rank r owns the contiguous block [r*N/G, (r+1)*N/G); the draw streams
are keyed by GLOBAL env index (include/salp_vec.h "Randomness"), so env i's trajectory is the
same for every G.  The only exchange is the all-gather of what the step returns
(`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" for the CPU tests); there is no
reduction on the data path.

`engine_factory(cfg, n_local, seed, env_index_base)` builds the per-rank simulator; the default
is the HIP `SalpVectorEnv`.  (tests/ inject a CPU checker engine to exercise the sharding and
gather logic under gloo — the product default never does.)
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist

from .vector_env import SalpVectorEnv, _as_config


def _to_tensor(x, device):
    if isinstance(x, torch.Tensor):
        return x
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


class ShardedSalpVectorEnv:
    def __init__(self, config="single_food", num_envs: int = 8 * 131072, *, process_group=None,
                 device: Optional[str] = None, seed: int = 0, engine_factory: Optional[Callable] = None,
                 gather_final_observation: bool = True, rehearse_shard: Optional[tuple] = None, **overrides):
        """rehearse_shard=(r, G): simulate shard r of a G-way split of `num_envs` on THIS process whatever its rank
        (envs [r N/G, (r+1) N/G), same global-index draw keys), with the collectives running over the real group — how a
        one-GPU box runs the per-GPU workload of a larger job (tests/test_gpu_sharded_workload.py: BASELINE configs[3],
        shard 7 of 8).  A real job never passes it."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (backend 'nccl' on GPUs)")
        self.pg = process_group
        self.rank = dist.get_rank(self.pg)
        self.world = dist.get_world_size(self.pg)
        shard, shards = (self.rank, self.world) if rehearse_shard is None else (int(rehearse_shard[0]), int(rehearse_shard[1]))
        if not 0 <= shard < shards:
            raise ValueError(f"rehearse_shard={rehearse_shard}: shard index out of range")
        if num_envs % shards != 0:
            raise ValueError(f"num_envs={num_envs} must be divisible by the number of shards {shards}")
        self.cfg = _as_config(config, **overrides)
        self.num_envs = int(num_envs)
        self.local_envs = self.num_envs // shards
        self.env_index_base = shard * self.local_envs
        if engine_factory is None:
            dev = device or f"cuda:{torch.cuda.current_device()}"
            self.engine = SalpVectorEnv(self.cfg, self.local_envs, device=dev, seed=seed,
                                        env_index_base=self.env_index_base)
            self.device = self.engine.device
        else:
            self.engine = engine_factory(self.cfg, self.local_envs, seed, self.env_index_base)
            self.device = torch.device("cpu")
        self.obs_dim, self.act_dim = self.cfg.obs_dim, self.cfg.act_dim
        self.gather_final_observation = bool(gather_final_observation)
        # all_gather_into_tensor is the one flat collective RCCL runs; a backend without it (older gloo) is
        # found ONCE here, by its NotImplementedError on a probe — never by catching errors of a real exchange
        # (an RCCL failure must surface, not be retried on a poisoned communicator).
        self._flat_gather = True
        if dist.get_backend(self.pg) != "nccl":
            probe_in = torch.zeros(1, device=self.device)
            probe_out = torch.zeros(self.world, device=self.device)
            try:
                dist.all_gather_into_tensor(probe_out, probe_in, group=self.pg)
            except NotImplementedError:
                self._flat_gather = False
        self._pending = []
        self._gbuf = {}
        self._pack = None
        self._slot, self._slot_work, self._stage = 0, [None, None], [None, None]

    # ------------------------------------------------------------------ collectives
    def _out(self, name, local: torch.Tensor):
        shape = (self.world * local.shape[0],) + tuple(local.shape[1:])
        b = self._gbuf.get(name)
        if b is None or tuple(b.shape) != shape or b.dtype != local.dtype:
            b = torch.empty(shape, dtype=local.dtype, device=local.device)
            self._gbuf[name] = b
        return b

    def all_gather(self, name: str, local, async_op: bool = False):
        """Concatenates every rank's block along dim 0 (rank order = env order)."""
        local = _to_tensor(local, self.device).contiguous()
        out = self._out(name, local)
        if self._flat_gather:
            work = dist.all_gather_into_tensor(out, local, group=self.pg, async_op=async_op)
        else:
            chunks = list(out.chunk(self.world, dim=0))
            work = dist.all_gather(chunks, local, group=self.pg, async_op=async_op)
        return (out, work) if async_op else out

    def _shard(self, actions, lead):
        """Accepts global [.., N, A] or local [.., N/G, A] actions; returns this rank's block."""
        if actions is None:
            return None
        n_axis = lead
        n = actions.shape[n_axis]
        if n == self.local_envs:
            return actions
        if n != self.num_envs:
            raise ValueError(f"actions have {n} envs; expected {self.num_envs} (global) or {self.local_envs} (local)")
        sl = [slice(None)] * actions.ndim
        sl[n_axis] = slice(self.env_index_base, self.env_index_base + self.local_envs)
        return actions[tuple(sl)]

    # ------------------------------------------------------------------ VectorEnv surface (global views)
    def reset(self, *, seed=None, options=None):
        obs, info = self.engine.reset(seed=seed, options=options)
        return self.all_gather("obs", obs), info

    def step(self, actions):
        """Every rank returns the full [N, ...] batch: obs, reward, terminated, truncated AND info — `info` holds
        global `food_collected`, `steps_since_food`, `collision` [N] and (unless the env was built with
        gather_final_observation=False) `final_observation` [N, obs_dim] with its mask `_final_observation`,
        so the consumer pattern of the single-GPU env (`torch.where(done[:, None], info["final_observation"],
        obs)`) bootstraps from the true terminal observation of every shard.  One collective per step: all of
        it travels as one [N/G, W] float32 block, W = obs_dim + 5 (+ obs_dim with terminal observations) — a
        ring all-gather over xGMI is latency-bound at this size, separate collectives cost one launch each.
        (The three info columns are small non-negative integers: exact in float32.)"""
        a = self._shard(actions, 0)
        obs, rew, term, trunc, info = self.engine.step(a)
        D = self.obs_dim
        obs, rew = _to_tensor(obs, self.device), _to_tensor(rew, self.device)
        flags = _to_tensor(term, self.device).to(torch.float32) + 2.0 * _to_tensor(trunc, self.device).to(torch.float32)
        fin = info.get("final_observation") if self.gather_final_observation else None
        W = D + 5 + (D if fin is not None else 0)
        pack = self._pack
        if pack is None or tuple(pack.shape) != (obs.shape[0], W) or pack.device != obs.device:
            pack = self._pack = torch.zeros((obs.shape[0], W), dtype=torch.float32, device=obs.device)
        pack[:, :D] = obs
        pack[:, D] = rew
        pack[:, D + 1] = flags
        for j, k in enumerate(("food_collected", "steps_since_food", "collision")):
            if k in info:
                pack[:, D + 2 + j] = _to_tensor(info[k], self.device).to(torch.float32)
        if fin is not None:
            pack[:, D + 5:] = _to_tensor(fin, self.device)
        g = self.all_gather("step", pack)
        gf = g[:, D + 1]
        g_term, g_trunc = (gf == 1.0) | (gf == 3.0), gf >= 2.0
        ginfo = {"food_collected": g[:, D + 2].to(torch.int32), "steps_since_food": g[:, D + 3].to(torch.int32),
                 "collision": g[:, D + 4].to(torch.int32), "local": info}
        if fin is not None:
            ginfo["final_observation"] = g[:, D + 5:]
            ginfo["_final_observation"] = g_term | g_trunc
        return g[:, :D], g[:, D], g_term, g_trunc, ginfo

    def rollout(self, actions=None, horizon=None, gather: str = "final", async_gather: bool = False):
        """Local fused rollout of `horizon` steps, then the exchange:
        gather="final": all-gather the last step's observation [N, obs_dim] (what a centralised
                        actor needs to continue), "all": the whole [H, N, obs_dim] block,
                        "none": no exchange (data-parallel learners).
        Returns (local_outputs, gathered_obs_or_None).  With async_gather the collective overlaps
        the caller's next launch; call wait_gather() before reading the gathered tensor."""
        if not (gather == "final" and async_gather):
            self.wait_gather()       # "all" reads the rollout's own output block: it must not be overwritten under it
        a = self._shard(actions, 1)
        out = self.engine.rollout(a, horizon)
        g = None
        if gather == "final":
            if async_gather:
                g = self.gather_final_async(out["obs"][-1])
            else:
                g = self.all_gather("final_obs", out["obs"][-1])
        elif gather == "all":
            H = out["obs"].shape[0]
            # [H, n, D] -> rank-major blocks; viewed back as [G, H, n, D] by the caller
            g = self.all_gather("all_obs", _to_tensor(out["obs"], self.device).reshape(1, H, self.local_envs, self.obs_dim),
                                async_op=async_gather)
            if async_gather:
                self._pending.append(g[1])
                g = g[0]
        elif gather != "none":
            raise ValueError("gather must be 'final', 'all' or 'none'")
        return out, g

    def gather_final_async(self, final_obs):
        """Asynchronous all-gather of one [N/G, obs_dim] block that does not pin the rollout's output buffer:
        the block is first copied (25 MB at N/G = 262144: ~10 us) into one of two staging buffers on the compute
        stream, and the collective reads the staging copy.  The next launch — which overwrites the rollout's
        output block — can therefore start at once, and the collective runs beside it on RCCL's own stream; a
        staging slot is only waited for when it comes round again, two launches later.  Returns the gathered
        tensor of this slot (valid after wait_gather())."""
        slot = self._slot
        self._slot ^= 1
        w = self._slot_work[slot]
        if w is not None:            # the collective that last used this slot (two launches ago)
            w.wait()
            self._slot_work[slot] = None
        local = _to_tensor(final_obs, self.device)
        st = self._stage[slot]
        if st is None or st.shape != local.shape or st.device != local.device:
            st = self._stage[slot] = torch.empty_like(local, memory_format=torch.contiguous_format)
        st.copy_(local)
        g, work = self.all_gather(f"final_obs{slot}", st, async_op=True)
        self._slot_work[slot] = work
        return g

    def wait_gather(self):
        for i, w in enumerate(self._slot_work):
            if w is not None:
                w.wait()
                self._slot_work[i] = None
        for w in self._pending:
            w.wait()
        self._pending = []

    def close(self):
        self.wait_gather()
        self.engine.close()
