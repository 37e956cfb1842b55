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
    def __init__(self, obs_dim: int, action_dim: int, hidden_sizes: Sequence[int] = (256, 256),
                 learning_rate: float = 3e-4, device="cuda"):
        super().__init__()
        self.net = mlp([obs_dim + action_dim, *hidden_sizes], 1)
        self.device = torch.device(device)
        self.to(self.device)
        self.optimizer = torch.optim.Adam(self.parameters(), lr=learning_rate)
        self.training_step = 0

    def logits(self, obs, action):
        return self.net(torch.cat([obs, action], dim=-1))

    def forward(self, obs, action):
        """P((obs, action) came from the expert), shape [B, 1] (discriminator.py:43-63)."""
        return torch.sigmoid(self.logits(obs, action))

    @torch.no_grad()
    def predict_reward(self, obs, action):
        """−log(1 − D(s,a) + 1e-8) (discriminator.py:65-85), shape [B]."""
        return -torch.log(1 - self.forward(obs, action) + 1e-8).squeeze(-1)

    def update(self, expert_batch: Dict[str, torch.Tensor], agent_batch: Dict[str, torch.Tensor]) -> Dict[str, float]:
        """One BCE step, expert label 1 / agent label 0 (discriminator.py:87-139)."""
        el = self.logits(expert_batch["observations"], expert_batch["actions"])
        al = self.logits(agent_batch["observations"], agent_batch["actions"])
        expert_loss = F.binary_cross_entropy_with_logits(el, torch.ones_like(el))
        agent_loss = F.binary_cross_entropy_with_logits(al, torch.zeros_like(al))
        loss = expert_loss + agent_loss
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        self.training_step += 1
        with torch.no_grad():
            el, al = el.detach(), al.detach()
            acc = ((el > 0).float().mean() + (al <= 0).float().mean()) / 2
            ep, ap = torch.sigmoid(el).mean(), torch.sigmoid(al).mean()
        loss, expert_loss, agent_loss = loss.detach(), expert_loss.detach(), agent_loss.detach()
        return {"discriminator_loss": float(loss), "expert_loss": float(expert_loss), "agent_loss": float(agent_loss),
                "discriminator_accuracy": float(acc), "expert_prob_mean": float(ep), "agent_prob_mean": float(ap)}


class ExpertBuffer:
    """Expert (obs, action) pairs on the device; `sample` as expert_buffer.py:73-102."""

    def __init__(self, obs_dim: int, action_dim: int, device="cuda"):
        self.obs_dim, self.action_dim, self.device = obs_dim, action_dim, torch.device(device)
        self.observations = torch.empty((0, obs_dim), device=self.device)
        self.actions = torch.empty((0, action_dim), device=self.device)
        self.episodes = 0

    def add_episode(self, observations, actions, metadata: Optional[dict] = None):
        o = torch.as_tensor(np.asarray(observations, dtype=np.float32)).reshape(-1, self.obs_dim).to(self.device)
        a = torch.as_tensor(np.asarray(actions, dtype=np.float32)).reshape(-1, self.action_dim).to(self.device)
        if o.shape[0] != a.shape[0]:
            raise ValueError("observations and actions differ in length")
        self.observations = torch.cat([self.observations, o])
        self.actions = torch.cat([self.actions, a])
        self.episodes += 1

    def load_npz(self, path: str):
        """`observations` [T, obs_dim] and `act32` or `actions` [T, action_dim] from a .npz demo."""
        z = np.load(path, allow_pickle=False)
        act = z["act32"] if "act32" in z.files else z["actions"]
        self.add_episode(z["observations"], act)

    def load_directory(self, directory: str, pattern: str = "*.npz") -> int:
        """expert_buffer.py:148-187 `load_directory`, for .npz demos.  Returns episodes loaded."""
        import glob
        import os
        n = 0
        for p in sorted(glob.glob(os.path.join(directory, pattern))):
            self.load_npz(p)
            n += 1
        return n

    def sample(self, batch_size: int) -> Dict[str, torch.Tensor]:
        if len(self) == 0:
            raise ValueError("expert buffer is empty")
        idx = torch.randint(0, len(self), (batch_size,), device=self.device)
        return {"observations": self.observations[idx], "actions": self.actions[idx]}

    def __len__(self):
        return int(self.observations.shape[0])


def gail_reward_fn(disc: Discriminator, env_weight: float = 0.3, gail_weight: float = 0.7):
    """reward = 0.3·r_env + 0.7·r_gail (configs/sac_gail.yaml:44-45), for sac.train_sac(reward_fn=…)."""
    def fn(obs, act, env_reward):
        return env_weight * env_reward + gail_weight * disc.predict_reward(obs, act)
    return fn
