"""GAIL reward path on device tensors (configs/sac_gail.yaml:39-48).

Mirrors the reference's `Discriminator` (src/salp/agents/discriminator.py:16-139: MLP on
concat(obs, act) → logit, BCE with expert label 1 / agent label 0, reward −log(1 − D + 1e-8)) and
the sampling surface of `ExpertBuffer` (src/salp/training/expert_buffer.py:34-102) with everything
resident on the GPU.  Demonstrations are loaded from `.npz` (the reference stores `.pkl`; pickles
are not loaded here — convert with a tool that does not unpickle, e.g.
tests/golden/extract_human_demos.py's reader).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .sac import mlp


class Discriminator(nn.Module):
    """The reference's GAIL discriminator (src/salp/agents/discriminator.py:16-139 over core/base_agent.py:12-73:
    Linear/ReLU stack on concat(obs, act), sigmoid output), pinned to it by tests/golden/gail_discriminator.npz:
    same weights give the same `forward`, `predict_reward` and — one `update` later — the same losses, metrics
    and parameters (test_gail_parity.py).  Differences of form, not of value:
      * the loss is evaluated in logit space, `softplus(-/+logit)` clamped at 100 — what
        `F.binary_cross_entropy(sigmoid(logit), label)` computes (PyTorch clamps its logs at -100), equal to float32
        rounding for |logit| < ~15 and without the saturation of the probability beyond (where the reference's
        loss jumps to the clamp and its gradient vanishes);
      * `update` returns 0-dim device tensors (`metrics_to_host` converts them with ONE host sync): six `.item()`
        calls per update, as the reference makes, are six pipeline drains next to a hipGraph-captured SAC step;
      * parameters are named `net.<2i>` (nn.Sequential) where the reference has `layers.<i>`:
        `load_reference_state_dict` / `reference_state_dict` map between the two."""

    def __init__(self, obs_dim: int, action_dim: int, hidden_sizes: Sequence[int] = (256, 256),
                 learning_rate: float = 3e-4, device="cuda"):
        super().__init__()
        self.net = mlp([obs_dim + action_dim, *hidden_sizes], 1)
        self.device = torch.device(device)
        self.to(self.device)
        self.optimizer = torch.optim.Adam(self.parameters(), lr=learning_rate)
        self.training_step = 0

    # ---- reference checkpoint compatibility (BaseNetwork.layers = ModuleList of Linear)
    def load_reference_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Loads a state_dict of the reference Discriminator (`layers.<i>.weight/bias`)."""
        self.load_state_dict({f"net.{2 * int(k.split('.')[1])}.{k.split('.')[2]}": v for k, v in sd.items()})

    def reference_state_dict(self) -> Dict[str, torch.Tensor]:
        return {f"layers.{int(k.split('.')[1]) // 2}.{k.split('.')[2]}": v for k, v in self.state_dict().items()}

    def logits(self, obs, action):
        return self.net(torch.cat([obs, action], dim=-1))

    def forward(self, obs, action):
        """P((obs, action) came from the expert), shape [B, 1] (discriminator.py:43-63)."""
        return torch.sigmoid(self.logits(obs, action))

    @torch.no_grad()
    def predict_reward(self, obs, action):
        """-log(1 - D(s,a) + 1e-8), shape [B, 1] as in the reference (discriminator.py:65-85)."""
        return -torch.log(1 - self.forward(obs, action) + 1e-8)

    def update(self, expert_batch: Dict[str, torch.Tensor], agent_batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """One BCE step, expert label 1 / agent label 0 (discriminator.py:87-139).  Batches: dicts with
        'observations' / 'actions' (tensors on the device, or anything `torch.as_tensor` takes, as the reference
        accepts numpy).  Returns the reference's six metrics as 0-dim tensors on the device."""
        dev = self.device
        cvt = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev)
        el = self.logits(cvt(expert_batch["observations"]), cvt(expert_batch["actions"]))
        al = self.logits(cvt(agent_batch["observations"]), cvt(agent_batch["actions"]))
        # F.binary_cross_entropy(sigmoid(x), 1) = min(softplus(-x), 100); (.., 0) = min(softplus(x), 100)
        expert_loss = F.softplus(-el).clamp(max=100.0).mean()
        agent_loss = F.softplus(al).clamp(max=100.0).mean()
        loss = expert_loss + agent_loss
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        self.training_step += 1
        with torch.no_grad():
            ep, ap = torch.sigmoid(el), torch.sigmoid(al)
            acc = ((ep > 0.5).float().mean() + (ap <= 0.5).float().mean()) / 2.0
            return {"discriminator_loss": loss.detach(), "expert_loss": expert_loss.detach(), "agent_loss": agent_loss.detach(),
                    "discriminator_accuracy": acc, "expert_prob_mean": ep.mean(), "agent_prob_mean": ap.mean()}

    @staticmethod
    def metrics_to_host(metrics: Dict[str, torch.Tensor]) -> Dict[str, float]:
        """The reference's dict of Python floats, with one device-to-host transfer for all six values."""
        keys = list(metrics)
        vals = torch.stack([metrics[k].reshape(()) for k in keys]).tolist()
        return dict(zip(keys, vals))


class ExpertBuffer:
    """Expert transitions on the device; `sample` keeps the contract of expert_buffer.py:73-102: a dict with
    'observations' [B, obs_dim], 'actions' [B, action_dim], 'rewards' [B], 'next_observations' [B, obs_dim],
    'dones' [B], rows drawn uniformly with replacement over ALL stored transitions, the five arrays indexed by the
    same draw.  (`rewards` / `next_observations` / `dones` are zeros when an episode was added without them —
    the discriminator reads only the first two.)"""

    KEYS = ("observations", "actions", "rewards", "next_observations", "dones")

    def __init__(self, obs_dim: int, action_dim: int, device="cuda"):
        self.obs_dim, self.action_dim, self.device = obs_dim, action_dim, torch.device(device)
        self.observations = torch.empty((0, obs_dim), device=self.device)
        self.actions = torch.empty((0, action_dim), device=self.device)
        self.rewards = torch.empty((0,), device=self.device)
        self.next_observations = torch.empty((0, obs_dim), device=self.device)
        self.dones = torch.empty((0,), device=self.device)
        self.episodes = 0

    @property
    def num_transitions(self) -> int:
        return len(self)

    def add_episode(self, observations, actions=None, metadata: Optional[dict] = None, rewards=None,
                    next_observations=None, dones=None):
        """`add_episode(obs, act, ...)` or, as the reference (expert_buffer.py:34-71), `add_episode(episode_dict)`."""
        if isinstance(observations, dict):
            ep = observations
            observations, actions = ep["observations"], ep["actions"]
            rewards, next_observations, dones = ep.get("rewards"), ep.get("next_observations"), ep.get("dones")
        f32 = lambda x, shape: torch.as_tensor(np.asarray(x, dtype=np.float32)).reshape(shape).to(self.device)
        o, a = f32(observations, (-1, self.obs_dim)), f32(actions, (-1, self.action_dim))
        T = o.shape[0]
        if a.shape[0] != T:
            raise ValueError("observations and actions differ in length")
        r = f32(rewards, (-1,)) if rewards is not None else torch.zeros(T, device=self.device)
        no = f32(next_observations, (-1, self.obs_dim)) if next_observations is not None else torch.zeros_like(o)
        d = f32(dones, (-1,)) if dones is not None else torch.zeros(T, device=self.device)
        if not (r.shape[0] == no.shape[0] == d.shape[0] == T):
            raise ValueError("episode arrays differ in length")
        self.observations = torch.cat([self.observations, o])
        self.actions = torch.cat([self.actions, a])
        self.rewards = torch.cat([self.rewards, r])
        self.next_observations = torch.cat([self.next_observations, no])
        self.dones = torch.cat([self.dones, d])
        self.episodes += 1

    def load_npz(self, path: str):
        """`observations` [T, obs_dim] and `act32` or `actions` [T, action_dim] from a .npz demo."""
        z = np.load(path, allow_pickle=False)
        act = z["act32"] if "act32" in z.files else z["actions"]
        opt = {k: z[k] for k in ("rewards", "next_observations", "dones") if k in z.files}
        self.add_episode(z["observations"], act, **opt)

    def load_directory(self, directory: str, pattern: str = "*.npz") -> int:
        """expert_buffer.py:148-187 `load_directory`, for .npz demos.  Returns episodes loaded."""
        import glob
        import os
        n = 0
        for p in sorted(glob.glob(os.path.join(directory, pattern))):
            self.load_npz(p)
            n += 1
        return n

    def sample(self, batch_size: int, indices=None) -> Dict[str, torch.Tensor]:
        """`indices` (optional, [batch_size] ints) replaces the uniform draw — the hook the parity test uses to feed
        the reference's own `np.random.randint` draw."""
        if len(self) == 0:
            raise ValueError("expert buffer is empty")
        if indices is None:
            idx = torch.randint(0, len(self), (batch_size,), device=self.device)
        else:
            idx = torch.as_tensor(np.asarray(indices), dtype=torch.long, device=self.device)
            if idx.shape != (batch_size,):
                raise ValueError("indices must have shape [batch_size]")
        return {k: getattr(self, k)[idx] for k in self.KEYS}

    def __len__(self):
        return int(self.observations.shape[0])


def gail_reward_fn(disc: Discriminator, env_weight: float = 0.3, gail_weight: float = 0.7):
    """reward = 0.3·r_env + 0.7·r_gail (configs/sac_gail.yaml:44-45), for sac.train_sac(reward_fn=…)."""
    def fn(obs, act, env_reward):
        return env_weight * env_reward + gail_weight * disc.predict_reward(obs, act).squeeze(-1)
    return fn
