"""Soft Actor-Critic on PyTorch-ROCm, consuming the batched simulator's device tensors
(BASELINE.json configs[4]).

The reference trains with stable-baselines3 `SAC("MlpPolicy", env, ...)` (train.py:60-70;
src/salp/agents/sb3_sac_agent.py:66-90), which is not vendored and cannot keep up with thousands of
envs (per-env Python info dicts, row-wise replay inserts).  This module restates that learner with
SB3's MlpPolicy architecture and defaults — actor obs→256→256→(mean, log_std) with tanh squashing
and log_std clamped to [-20, 2]; twin critics (obs, act)→256→256→1, ReLU; Adam 3e-4; polyak tau;
automatic entropy tuning with target entropy −|A| unless the YAML's `alpha` / `target_entropy`
are given (sb3_sac_agent.py:77-79) — on tensors that never leave the GPU: the replay buffer is a
set of device tensors filled N rows per env step.

Pure PyTorch (the policy nets are the one place the north_star assigns to PyTorch-ROCm); the
environment stepping goes through the HIP library.
"""
from __future__ import annotations

import dataclasses
import math
import time
from typing import Dict, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

LOG_STD_MIN, LOG_STD_MAX = -20.0, 2.0   # stable_baselines3.sac.policies


def mlp(sizes: Sequence[int], out_dim: int) -> nn.Sequential:
    layers, d = [], sizes[0]
    for h in sizes[1:]:
        layers += [nn.Linear(d, h), nn.ReLU()]
        d = h
    layers.append(nn.Linear(d, out_dim))
    return nn.Sequential(*layers)


class Actor(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden=(256, 256), act_low=None, act_high=None):
        super().__init__()
        body, d = [], obs_dim
        for h in hidden:
            body += [nn.Linear(d, h), nn.ReLU()]
            d = h
        self.body = nn.Sequential(*body)
        self.mu = nn.Linear(d, act_dim)
        self.log_std = nn.Linear(d, act_dim)
        low = torch.full((act_dim,), -1.0) if act_low is None else torch.as_tensor(act_low, dtype=torch.float32)
        high = torch.full((act_dim,), 1.0) if act_high is None else torch.as_tensor(act_high, dtype=torch.float32)
        # SB3 squashes to [-1, 1] and rescales to the Box bounds (snake:69-74: [-1,1] or [0,1]x[-1,1])
        self.register_buffer("scale", (high - low) / 2)
        self.register_buffer("shift", (high + low) / 2)

    def forward(self, obs, deterministic=False, with_logprob=True):
        h = self.body(obs)
        mu, log_std = self.mu(h), self.log_std(h).clamp(LOG_STD_MIN, LOG_STD_MAX)
        std = log_std.exp()
        u = mu if deterministic else mu + std * torch.randn_like(mu)
        a = torch.tanh(u)
        logp = None
        if with_logprob:
            # log N(u; mu, std) - sum log(1 - tanh(u)^2), the numerically stable form
            logp = (-0.5 * ((u - mu) / std) ** 2 - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
            logp = logp - (2 * (math.log(2) - u - F.softplus(-2 * u))).sum(-1)
        return a * self.scale + self.shift, logp

    def unscale(self, action):
        return (action - self.shift) / self.scale


class TwinQ(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden=(256, 256)):
        super().__init__()
        self.q1 = mlp([obs_dim + act_dim, *hidden], 1)
        self.q2 = mlp([obs_dim + act_dim, *hidden], 1)

    def forward(self, obs, act):
        x = torch.cat([obs, act], dim=-1)
        return self.q1(x).squeeze(-1), self.q2(x).squeeze(-1)


class DeviceReplayBuffer:
    """Ring buffer of transitions as device tensors; `add` takes N rows at once."""

    def __init__(self, capacity, obs_dim, act_dim, device):
        self.capacity, self.device = int(capacity), device
        self.obs = torch.empty((self.capacity, obs_dim), device=device)
        self.next_obs = torch.empty((self.capacity, obs_dim), device=device)
        self.act = torch.empty((self.capacity, act_dim), device=device)
        self.rew = torch.empty((self.capacity,), device=device)
        self.term = torch.empty((self.capacity,), device=device)   # 1.0 where the episode TERMINATED (no bootstrap)
        self.pos, self.size = 0, 0
        # device-side copies of the cursor, used by the capturable variants below
        self.pos_t = torch.zeros((), dtype=torch.long, device=device)
        self.size_t = torch.zeros((), dtype=torch.long, device=device)

    # --- hipGraph-capturable variants: no Python-side cursor is baked into the captured kernels ---------
    @torch.no_grad()
    def add_capturable(self, obs, act, rew, next_obs, terminated):
        """Same as `add` for n <= capacity rows, as pure device ops on `pos_t` / `size_t` (the captured
        graph is replayed with a moving cursor).  Call `advance_host(n)` after each replay / call."""
        n = obs.shape[0]
        assert n <= self.capacity
        idx = (self.pos_t + torch.arange(n, device=self.device)) % self.capacity
        self.obs.index_copy_(0, idx, obs)
        self.next_obs.index_copy_(0, idx, next_obs)
        self.act.index_copy_(0, idx, act)
        self.rew.index_copy_(0, idx, rew)
        self.term.index_copy_(0, idx, terminated.to(self.rew.dtype))
        self.pos_t.add_(n).remainder_(self.capacity)
        self.size_t.add_(n).clamp_(max=self.capacity)

    def advance_host(self, n):
        self.pos = (self.pos + n) % self.capacity
        self.size = min(self.capacity, self.size + n)

    def sample_capturable(self, batch_size):
        idx = (torch.rand(batch_size, device=self.device) * self.size_t).long().clamp_(max=self.capacity - 1)
        return self.obs[idx], self.act[idx], self.rew[idx], self.next_obs[idx], self.term[idx]

    @torch.no_grad()
    def add(self, obs, act, rew, next_obs, terminated):
        n = obs.shape[0]
        if n >= self.capacity:
            obs, act, rew, next_obs, terminated = (t[-self.capacity:] for t in (obs, act, rew, next_obs, terminated))
            n = self.capacity
        end = self.pos + n
        if end <= self.capacity:
            sl = slice(self.pos, end)
            self.obs[sl], self.act[sl], self.rew[sl], self.next_obs[sl] = obs, act, rew, next_obs
            self.term[sl] = terminated.to(self.rew.dtype)
        else:
            k = self.capacity - self.pos
            self.add(obs[:k], act[:k], rew[:k], next_obs[:k], terminated[:k])
            self.add(obs[k:], act[k:], rew[k:], next_obs[k:], terminated[k:])
            return
        self.pos = end % self.capacity
        self.size = min(self.capacity, self.size + n)
        self.pos_t.fill_(self.pos)
        self.size_t.fill_(self.size)

    def sample(self, batch_size, generator=None):
        idx = torch.randint(0, self.size, (batch_size,), device=self.device, generator=generator)
        return self.obs[idx], self.act[idx], self.rew[idx], self.next_obs[idx], self.term[idx]


@dataclasses.dataclass
class SACConfig:
    """`agent:` block of the reference's YAML presets (configs/*.yaml)."""
    hidden_sizes: Sequence[int] = (256, 256)
    learning_rate: float = 3e-4
    batch_size: int = 128
    buffer_size: int = 500_000
    gamma: float = 0.99
    tau: float = 0.005
    alpha: Optional[float] = None            # None = learned ("auto", SB3 default); else fixed
    target_entropy: Optional[float] = None   # None = -act_dim
    alpha_lr: float = 3e-4
    learning_starts: int = 1000              # env-steps (per env) of random actions before updates
    updates_per_step: int = 1                # gradient steps per vector env step

    @classmethod
    def from_preset(cls, name: str) -> "SACConfig":
        p = AGENT_PRESETS[name]
        return cls(**p)


# `agent:` blocks of configs/single_food.yaml:20-33, single_food_long_horizon.yaml:20-33, sac_gail.yaml:16-29
AGENT_PRESETS: Dict[str, dict] = {
    # src/salp/environments/train_robot.py:39-47 (the HEAD simulator's SAC: ent_coef "auto", batch 512, buffer 1e5)
    "salp_robot": dict(batch_size=512, buffer_size=100_000, gamma=0.99, tau=0.005, alpha=None, target_entropy=None),
    "single_food": dict(batch_size=128, buffer_size=500_000, gamma=0.99, tau=0.005, alpha=0.5, target_entropy=-0.5),
    "single_food_long_horizon": dict(batch_size=256, buffer_size=100_000, gamma=0.995, tau=0.005, alpha=0.2,
                                     target_entropy=-0.5),
    "sac_gail": dict(batch_size=128, buffer_size=500_000, gamma=0.99, tau=0.005, alpha=0.1, target_entropy=-1.0),
}


class SAC:
    def __init__(self, obs_dim, act_dim, cfg: SACConfig = SACConfig(), device="cuda", act_low=None, act_high=None,
                 learn_alpha: Optional[bool] = None, seed: Optional[int] = None, process_group=None,
                 data_parallel: bool = False):
        """`data_parallel=True` (needs an initialised `torch.distributed`, backend "nccl" = RCCL on GPUs):
        one learner replica per rank, each fed by its own env shard and replay buffer; parameters are
        broadcast from rank 0 at construction and every backward is followed by one flat all-reduce
        (mean) of that network's gradients, so the replicas stay bit-identical."""
        self.cfg, self.device = cfg, torch.device(device)
        self.pg, self.data_parallel = process_group, bool(data_parallel)
        if seed is not None:
            torch.manual_seed(seed)
        self.actor = Actor(obs_dim, act_dim, cfg.hidden_sizes, act_low, act_high).to(self.device)
        self.critic = TwinQ(obs_dim, act_dim, cfg.hidden_sizes).to(self.device)
        self.critic_target = TwinQ(obs_dim, act_dim, cfg.hidden_sizes).to(self.device)
        self.critic_target.load_state_dict(self.critic.state_dict())
        for p in self.critic_target.parameters():
            p.requires_grad_(False)
        # capturable Adam keeps its step count on the device, so `update` can live inside a hipGraph
        # (and fused: one multi-tensor kernel per optimiser step instead of a dozen small ones per parameter)
        cap = self.device.type == "cuda"
        kw = dict(capturable=True, fused=True) if cap else {}
        self.actor_opt = torch.optim.Adam(self.actor.parameters(), lr=cfg.learning_rate, **kw)
        self.critic_opt = torch.optim.Adam(self.critic.parameters(), lr=cfg.learning_rate, **kw)
        # train.py:60-70 leaves ent_coef at SB3's "auto"; SB3SACAgent passes the YAML alpha (fixed)
        self.learn_alpha = (cfg.alpha is None) if learn_alpha is None else learn_alpha
        init_alpha = 1.0 if cfg.alpha is None else float(cfg.alpha)
        self.log_alpha = torch.tensor(math.log(init_alpha), device=self.device, requires_grad=self.learn_alpha)
        self.alpha_opt = torch.optim.Adam([self.log_alpha], lr=cfg.alpha_lr, **kw) if self.learn_alpha else None
        self.target_entropy = -float(act_dim) if cfg.target_entropy is None else float(cfg.target_entropy)
        self.updates = 0
        self.world = 1
        self.exchange = False      # gradients go through the flat exchange buffers (world > 1, or forced by a test)
        self.keep_grads = False    # zero the .grad tensors in place instead of dropping them (segmented graphs, see train_sac_graphed)
        if self.data_parallel:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("data_parallel=True needs torch.distributed to be initialised")
            self.world = dist.get_world_size(self.pg)
            src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0
            with torch.no_grad():
                for t in [*self.actor.parameters(), *self.critic.parameters(), *self.critic_target.parameters(),
                          self.log_alpha]:
                    dist.broadcast(t, src=src, group=self.pg)
            if seed is not None:    # same weights everywhere, different exploration noise per replica
                torch.manual_seed(seed + 1 + dist.get_rank(self.pg))
            self.exchange = self.world > 1

    # ------------------------------------------------------------------ one update = three capturable stages
    # An update has two points where data-parallel replicas must exchange gradients (after the critic's backward,
    # after the actor's / entropy coefficient's).  It is therefore written as three stages with NO collective
    # inside: `update()` runs them back to back with the all-reduces in between, and `train_sac_graphed` captures
    # each stage into its own hipGraph segment and issues the all-reduces between the replays — the whole
    # iteration stays graph-replayed on every rank (at world size 1 the three stages are one graph).
    # What crosses a stage boundary lives in persistent tensors (`_flat_*`, `_carry`), never in autograd state.
    def _flat_for(self, name, params):
        n = sum(p.numel() for p in params)
        f = getattr(self, name, None)
        if f is None or f.numel() != n:
            f = torch.zeros(n, device=self.device)
            setattr(self, name, f)
        return f

    @staticmethod
    def _pack(flat, params):
        torch.cat([p.grad.reshape(-1) for p in params], out=flat)

    @staticmethod
    def _unpack(flat, params):
        o = 0
        for p in params:
            n = p.numel()
            p.grad.copy_(flat[o:o + n].view_as(p))
            o += n

    def _all_reduce_mean(self, flat):
        """One all-reduce for a whole network (a few hundred KB: a single bucket is the right size for xGMI's
        per-link ring), then the mean.  Eager: called between graph segments."""
        if self.world == 1:
            return
        import torch.distributed as dist
        dist.all_reduce(flat, group=self.pg)
        flat.div_(self.world)

    def _carry_set(self, **kw):
        c = self.__dict__.setdefault("_carry", {})
        for k, v in kw.items():
            if k in c and c[k].shape == v.shape:
                c[k].copy_(v.detach())       # values only: nothing of the autograd graph crosses a stage
            else:
                c[k] = v.detach().clone()

    def stage_critic(self, batch):
        """Twin-critic loss and backward; leaves the flat critic gradient in `_flat_c`."""
        obs, act, rew, next_obs, term = batch
        alpha = self.log_alpha.exp().detach()
        with torch.no_grad():
            na, nlogp = self.actor(next_obs)
            tq1, tq2 = self.critic_target(next_obs, na)
            target = rew + self.cfg.gamma * (1.0 - term) * (torch.min(tq1, tq2) - alpha * nlogp)
        q1, q2 = self.critic(obs, act)
        critic_loss = 0.5 * (F.mse_loss(q1, target) + F.mse_loss(q2, target))
        self.critic_opt.zero_grad(set_to_none=not self.keep_grads)
        critic_loss.backward()
        self._carry_set(obs=obs, critic_loss=critic_loss)
        if self.exchange:
            self._cparams = list(self.critic.parameters())
            self._pack(self._flat_for("_flat_c", self._cparams), self._cparams)

    def stage_actor(self):
        """Critic step (with the averaged gradient), then actor / entropy-coefficient losses and backward; leaves
        their flat gradient in `_flat_a`."""
        if self.exchange:
            self._unpack(self._flat_c, self._cparams)
        self.critic_opt.step()
        obs = self._carry["obs"]
        alpha = self.log_alpha.exp().detach()
        pa, logp = self.actor(obs)
        pq1, pq2 = self.critic(obs, pa)
        actor_loss = (alpha * logp - torch.min(pq1, pq2)).mean()
        self.actor_opt.zero_grad(set_to_none=not self.keep_grads)
        actor_loss.backward()
        if self.learn_alpha:
            alpha_loss = -(self.log_alpha * (logp.detach() + self.target_entropy).mean())
            self.alpha_opt.zero_grad(set_to_none=not self.keep_grads)
            alpha_loss.backward()
        self._carry_set(actor_loss=actor_loss, entropy=-logp.detach().mean())
        if self.exchange:
            self._aparams = list(self.actor.parameters()) + ([self.log_alpha] if self.learn_alpha else [])
            self._pack(self._flat_for("_flat_a", self._aparams), self._aparams)

    def stage_finish(self) -> Dict[str, torch.Tensor]:
        """Actor and entropy-coefficient steps, polyak update of the target critic."""
        if self.exchange:
            self._unpack(self._flat_a, self._aparams)
        self.actor_opt.step()
        if self.learn_alpha:
            self.alpha_opt.step()
        with torch.no_grad():  # polyak update (core/base_agent.py:63-74 soft_update), two multi-tensor kernels
            tps, ps = list(self.critic_target.parameters()), list(self.critic.parameters())
            torch._foreach_mul_(tps, 1.0 - self.cfg.tau)
            torch._foreach_add_(tps, ps, alpha=self.cfg.tau)
        c = self._carry
        return {"critic_loss": c["critic_loss"], "actor_loss": c["actor_loss"],
                "alpha": self.log_alpha.exp().detach(), "entropy": c["entropy"]}

    @torch.no_grad()
    def act(self, obs, deterministic=False):
        a, _ = self.actor(obs, deterministic=deterministic, with_logprob=False)
        return a

    def update(self, batch) -> Dict[str, torch.Tensor]:
        self.stage_critic(batch)
        if self.exchange:
            self._all_reduce_mean(self._flat_c)
        self.stage_actor()
        if self.exchange:
            self._all_reduce_mean(self._flat_a)
        out = self.stage_finish()
        self.updates += 1
        return out

    def state_dict(self):
        return {"actor": self.actor.state_dict(), "critic": self.critic.state_dict(),
                "critic_target": self.critic_target.state_dict(), "log_alpha": self.log_alpha.detach().clone()}

    def load_state_dict(self, sd):
        self.actor.load_state_dict(sd["actor"])
        self.critic.load_state_dict(sd["critic"])
        self.critic_target.load_state_dict(sd["critic_target"])
        with torch.no_grad():
            self.log_alpha.copy_(sd["log_alpha"])


def train_sac(env, agent: SAC, total_vector_steps: int, buffer: Optional[DeviceReplayBuffer] = None,
              reward_fn=None, log_every: int = 0, stop_at_first_food: bool = False) -> Dict[str, float]:
    """Collect with the vector env, learn from the device replay buffer.

    env: SalpVectorEnv(output="torch") (or anything with its step/reset surface returning device
    tensors).  `reward_fn(obs, act, env_reward)` lets a GAIL discriminator mix its reward in
    (configs/sac_gail.yaml:44-45).  Returns timing / progress metrics, including the wall-clock to
    the first food capture of any env (BASELINE.json configs[4])."""
    cfg, dev = agent.cfg, agent.device
    n = env.num_envs
    buffer = buffer or DeviceReplayBuffer(cfg.buffer_size, env.obs_dim, env.act_dim, dev)
    low = torch.as_tensor(env.single_action_space.low, device=dev)
    high = torch.as_tensor(env.single_action_space.high, device=dev)
    obs, _ = env.reset()
    obs = obs.clone()
    t0 = time.perf_counter()
    first_food_s, first_food_step = None, None
    ep_returns = torch.zeros(n, device=dev)
    finished_returns, finished = 0.0, 0
    last = {}
    for step in range(total_vector_steps):
        if step < cfg.learning_starts:
            act = low + (high - low) * torch.rand((n, env.act_dim), device=dev)
        else:
            act = agent.act(obs)
        nobs, rew, term, trunc, info = env.step(act)
        done = term | trunc
        next_obs = torch.where(done[:, None], info["final_observation"], nobs)
        r = rew if reward_fn is None else reward_fn(obs, act, rew)
        buffer.add(obs, act, r, next_obs, term)
        ep_returns += rew
        if done.any():
            finished_returns += float(ep_returns[done].sum())
            finished += int(done.sum())
            ep_returns[done] = 0.0
        # "first food capture" for SalpSnakeEnv; for envs without a food counter (the HEAD SalpRobotEnv) the first
        # termination, i.e. the first robot that reaches its target (salp_robot_env.py:176-178)
        got_food = (info["food_collected"] > 0).any() if "food_collected" in info else term.any()
        if agent.world > 1 and first_food_s is None:   # every rank must take the same branch (the updates are collective)
            import torch.distributed as dist
            flag = got_food.to(torch.int32).reshape(1)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=agent.pg)
            got_food = flag[0] > 0
        if first_food_s is None and bool(got_food):
            torch.cuda.synchronize() if dev.type == "cuda" else None
            first_food_s, first_food_step = time.perf_counter() - t0, step + 1
            if stop_at_first_food:
                obs = nobs.clone()
                break
        obs = nobs.clone()
        if step >= cfg.learning_starts and buffer.size >= cfg.batch_size:
            for _ in range(cfg.updates_per_step):
                last = agent.update(buffer.sample(cfg.batch_size))
        if log_every and (step + 1) % log_every == 0:
            msg = {k: float(v) for k, v in last.items()}
            print(f"[sac] step {step + 1} env-steps {(step + 1) * n} updates {agent.updates} "
                  f"episodes {finished} mean return {finished_returns / max(finished, 1):.2f} {msg}", flush=True)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return {"wall_s": wall, "vector_steps": step + 1, "env_steps": (step + 1) * n, "updates": agent.updates,
            "first_food_wall_s": first_food_s, "first_food_vector_step": first_food_step,
            "episodes": finished, "mean_return": finished_returns / max(finished, 1),
            **{k: float(v) for k, v in last.items()}}


def train_sac_graphed(env, agent: SAC, total_vector_steps: int, buffer: Optional[DeviceReplayBuffer] = None,
                      reward_fn=None, stop_at_first_food: bool = False, poll_every: int = 8,
                      warmup_iters: int = 3, force_segments: bool = False) -> Dict[str, float]:
    """`train_sac` with hipGraph replays instead of ~250 kernel launches per vector step.

    One training iteration is ~250 small kernels (policy forward, `salp_vec_step`, replay insert, twin-critic
    / actor / entropy updates, polyak): issued one by one the host is the bottleneck (~5 ms per iteration at
    4096 envs, the GPU idle most of the time).  Here the whole iteration — acting, the env step through the
    C ABI, the buffer insert with a device-side cursor, sampling, the three optimiser steps — is captured
    once per phase (random actions before `learning_starts`, policy + updates after) and replayed; the host
    only polls a device flag every `poll_every` steps.  Same algorithm and hyper-parameters as `train_sac`;
    `reward_fn`, if given, must be capturable (pure device ops, no host reads).

    Data-parallel replicas (`SAC(data_parallel=True)`, one process per GPU): a collective cannot sit inside the
    captured region here, so the iteration is captured as SEGMENTS that end where gradients must be exchanged
    (`SAC.stage_critic / stage_actor / stage_finish`): [act + env step + insert + sample + critic backward]
    all-reduce [critic step + actor backward] all-reduce [actor step + polyak (+ next update's critic backward)]…
    — 1 + 2 x updates_per_step replays and 2 x updates_per_step small all-reduces per vector step, every rank
    replaying the same sequence.  `force_segments=True` uses the segmented form at world size 1 (tests)."""
    cfg, dev = agent.cfg, agent.device
    if dev.type != "cuda":
        raise RuntimeError("train_sac_graphed needs a ROCm device")
    segmented = agent.world > 1 or force_segments
    saved_mode = (agent.exchange, agent.keep_grads)
    agent.exchange = segmented
    try:
        return _train_sac_graphed(env, agent, total_vector_steps, buffer, reward_fn, stop_at_first_food, poll_every,
                                  warmup_iters, segmented)
    finally:
        agent.exchange, agent.keep_grads = saved_mode     # the agent leaves as it came (round 2 left exchange switched on)


def _train_sac_graphed(env, agent, total_vector_steps, buffer, reward_fn, stop_at_first_food, poll_every, warmup_iters,
                       segmented) -> Dict[str, float]:
    cfg, dev = agent.cfg, agent.device
    n = env.num_envs
    buffer = buffer or DeviceReplayBuffer(cfg.buffer_size, env.obs_dim, env.act_dim, dev)
    low = torch.as_tensor(env.single_action_space.low, device=dev)
    high = torch.as_tensor(env.single_action_space.high, device=dev)
    obs0, _ = env.reset()
    obs = obs0.clone()                                   # static input of every graph
    ep_returns = torch.zeros(n, device=dev)
    fin_ret = torch.zeros((), device=dev, dtype=torch.float64)
    fin_cnt = torch.zeros((), device=dev, dtype=torch.long)
    step_t = torch.zeros((), device=dev, dtype=torch.long)
    first_t = torch.full((), -1, device=dev, dtype=torch.long)
    last: Dict[str, torch.Tensor] = {}

    def seg_act(random_actions: bool, learn: bool):
        with torch.no_grad():
            if random_actions:
                act = low + (high - low) * torch.rand((n, env.act_dim), device=dev)
            else:
                act = agent.act(obs)
            nobs, rew, term, trunc, info = env.step(act)
            done = term | trunc
            next_obs = torch.where(done[:, None], info["final_observation"], nobs)
            r = rew if reward_fn is None else reward_fn(obs, act, rew)
            buffer.add_capturable(obs, act, r, next_obs, term)
            ep_returns.add_(rew)
            fin_ret.add_((ep_returns * done).sum())
            fin_cnt.add_(done.sum())
            ep_returns.mul_(~done)
            step_t.add_(1)
            hit = (info["food_collected"] > 0).any() if "food_collected" in info else term.any()
            first_t.copy_(torch.where((first_t < 0) & hit, step_t, first_t))
            obs.copy_(nobs)
        if learn:
            agent.stage_critic(buffer.sample_capturable(cfg.batch_size))

    def seg_finish(more: bool):
        out = agent.stage_finish()
        for k, v in out.items():
            if k in last:
                last[k].copy_(v)
            else:
                last[k] = v.clone()
        if more:
            agent.stage_critic(buffer.sample_capturable(cfg.batch_size))

    def plan(random_actions: bool, learn: bool):
        """The iteration as a list of ("run", fn) / ("reduce", flat-name) items; a new graph segment starts after
        every reduce.  Unsegmented: the same functions with nothing between them (one graph)."""
        items = [("run", lambda: seg_act(random_actions, learn))]
        if learn:
            ups = cfg.updates_per_step
            for u in range(ups):
                items += [("reduce", "_flat_c"), ("run", agent.stage_actor), ("reduce", "_flat_a"),
                          ("run", (lambda more: (lambda: seg_finish(more)))(u + 1 < ups))]
        if not segmented:
            items = [it for it in items if it[0] == "run"]
        return items

    def run_eager(items):
        for kind, x in items:
            if kind == "run":
                x()
            else:
                agent._all_reduce_mean(getattr(agent, x))

    def capture(items):
        """[(graph | None, reduce-name | None)]: consecutive "run" items share one graph segment."""
        segs, cur = [], []
        for kind, x in items + [("reduce", None)]:
            if kind == "run":
                cur.append(x)
                continue
            if cur:
                g = torch.cuda.CUDAGraph()
                fns = list(cur)
                with torch.cuda.graph(g, pool=pool):
                    for f in fns:
                        f()
                segs.append((g, x))
                cur = []
            elif x is not None:
                segs.append((None, x))
        return segs

    pool = torch.cuda.graph_pool_handle()
    graphs: Dict[tuple, list] = {}
    warm: Dict[tuple, int] = {}
    side = torch.cuda.Stream(device=dev)
    t0 = time.perf_counter()
    first_food_s, first_food_step = None, None
    t_learn, learn_from = None, 0
    step = 0
    while step < total_vector_steps:
        random_actions = step < cfg.learning_starts
        learn = (not random_actions) and buffer.size >= cfg.batch_size
        key = (random_actions, learn)
        if key in graphs:
            if learn and t_learn is None:
                torch.cuda.synchronize(dev)
                t_learn, learn_from = time.perf_counter(), step
            for g, red in graphs[key]:
                if g is not None:
                    g.replay()
                if red is not None:
                    agent._all_reduce_mean(getattr(agent, red))
            if learn:
                agent.updates += cfg.updates_per_step
        elif warm.get(key, 0) < warmup_iters:
            # a few eager iterations first (they are real training steps): allocator pools, autograd and
            # optimiser state must exist before capture
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                run_eager(plan(random_actions, learn))
            torch.cuda.current_stream(dev).wait_stream(side)
            if learn:
                agent.updates += cfg.updates_per_step
            warm[key] = warm.get(key, 0) + 1
        else:
            torch.cuda.synchronize(dev)
            if segmented and learn:
                # Segments of one iteration read and write the same .grad tensors (backward in one, _unpack / optimiser
                # step in the next).  Keep the tensors the eager warm-up iterations allocated (ordinary memory, stable
                # addresses) and zero them in place, instead of letting the first captured backward allocate them in the
                # graph pool, where their lifetime across segments rested on replay order alone.
                agent.keep_grads = True
            graphs[key] = capture(plan(random_actions, learn))      # capture runs no kernel
            continue
        step += 1
        buffer.advance_host(n)
        if first_food_s is None and (step % poll_every == 0 or step == total_vector_steps):
            if agent.world > 1:      # every rank must leave the loop at the same step: earliest capture of any replica
                import torch.distributed as dist
                f_t = torch.where(first_t < 0, torch.full_like(first_t, 1 << 60), first_t)
                dist.all_reduce(f_t, op=dist.ReduceOp.MIN, group=agent.pg)
                f = int(f_t.item())
                f = -1 if f >= (1 << 60) else f
            else:
                f = int(first_t.item())                  # the only host read of the loop
            if f >= 0:
                first_food_s, first_food_step = time.perf_counter() - t0, f
                if stop_at_first_food:
                    break
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    finished = int(fin_cnt.item())
    return {"wall_s": wall, "vector_steps": step, "env_steps": step * n, "updates": agent.updates,
            "first_food_wall_s": first_food_s, "first_food_vector_step": first_food_step,
            "episodes": finished, "mean_return": float(fin_ret.item()) / max(finished, 1),
            "graphs": sum(1 for segs in graphs.values() for g, _ in segs if g is not None),
            "segmented": segmented,
            "learn_ms_per_vector_step": None if t_learn is None or step <= learn_from else
            (t0 + wall - t_learn) * 1e3 / (step - learn_from),
            **{k: float(v) for k, v in last.items()}}
