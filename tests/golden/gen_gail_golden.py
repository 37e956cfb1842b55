#!/usr/bin/env python3
"""Generates tests/golden/gail_discriminator.npz and gail_expert_buffer.npz by running the REFERENCE's own GAIL
classes in the build container (torch + numpy only, CPU):
    src/salp/agents/discriminator.py  Discriminator.forward / predict_reward / update   (:43-139)
    src/salp/core/base_agent.py       BaseNetwork                                       (:12-73)
    src/salp/training/expert_buffer.py ExpertBuffer.add_episode / sample                (:34-102)
The files are loaded by path under the module names they import each other by; nothing of the reference is
stored — the fixtures hold inputs, the initial weights and what the reference computed from them.
Run:  python tests/golden/gen_gail_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SALP_REFERENCE", "/root/reference")


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def reference_classes():
    for pkg in ("salp", "salp.core", "salp.agents", "salp.training"):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    load("salp.core.base_agent", "src/salp/core/base_agent.py")
    disc = load("salp.agents.discriminator", "src/salp/agents/discriminator.py")
    eb = load("salp.training.expert_buffer", "src/salp/training/expert_buffer.py")
    return disc.Discriminator, eb.ExpertBuffer


def main():
    Discriminator, ExpertBuffer = reference_classes()
    obs_dim, act_dim, hidden, lr, B = 24, 1, [64, 48], 3e-4, 96
    torch.manual_seed(20260104)
    d = Discriminator(obs_dim, act_dim, hidden, activation="relu", learning_rate=lr, device="cpu")
    w0 = {k: v.detach().clone().numpy() for k, v in d.state_dict().items()}
    g = np.random.default_rng(7)
    obs = g.normal(0, 1, (B, obs_dim)).astype(np.float32)
    act = g.uniform(-1, 1, (B, act_dim)).astype(np.float32)
    with torch.no_grad():
        prob = d.forward(torch.from_numpy(obs), torch.from_numpy(act)).numpy()
    rew = d.predict_reward(torch.from_numpy(obs), torch.from_numpy(act)).numpy()
    expert = {"observations": g.normal(0.5, 1, (B, obs_dim)).astype(np.float32), "actions": g.uniform(-1, 1, (B, act_dim)).astype(np.float32)}
    agent = {"observations": g.normal(-0.5, 1, (B, obs_dim)).astype(np.float32), "actions": g.uniform(-1, 1, (B, act_dim)).astype(np.float32)}
    metrics = [d.update(expert, agent) for _ in range(3)]          # three Adam steps on the same batches
    w3 = {k: v.detach().clone().numpy() for k, v in d.state_dict().items()}
    keys = sorted(metrics[0])
    np.savez(os.path.join(HERE, "gail_discriminator.npz"),
             meta=np.array([obs_dim, act_dim, B, 3] + hidden), lr=np.float64(lr),
             obs=obs, act=act, prob=prob, reward=rew,
             expert_obs=expert["observations"], expert_act=expert["actions"],
             agent_obs=agent["observations"], agent_act=agent["actions"],
             metric_keys=np.array(keys), metrics=np.array([[m[k] for k in keys] for m in metrics], np.float64),
             **{"w0_" + k: v for k, v in w0.items()}, **{"w3_" + k: v for k, v in w3.items()})

    # ExpertBuffer: two episodes, then the reference's own draw (np.random.randint under a fixed seed)
    eb = ExpertBuffer(obs_dim, act_dim)
    eps = []
    for T in (40, 25):
        ep = {"observations": g.normal(0, 1, (T, obs_dim)).astype(np.float32), "actions": g.uniform(-1, 1, (T, act_dim)).astype(np.float32),
              "rewards": g.normal(0, 1, T).astype(np.float32), "next_observations": g.normal(0, 1, (T, obs_dim)).astype(np.float32),
              "dones": (g.uniform(0, 1, T) > 0.9).astype(np.float32)}
        eb.add_episode(ep)
        eps.append(ep)
    np.random.seed(11)
    idx = np.random.randint(0, eb.num_transitions, size=32)        # the draw ExpertBuffer.sample makes (:93)
    np.random.seed(11)
    s = eb.sample(32)
    np.savez(os.path.join(HERE, "gail_expert_buffer.npz"), indices=idx, num_transitions=np.int64(eb.num_transitions),
             **{f"ep{i}_{k}": v for i, ep in enumerate(eps) for k, v in ep.items()},
             **{"sample_" + k: v for k, v in s.items()})
    print("wrote gail_discriminator.npz, gail_expert_buffer.npz;", "loss", metrics[0]["discriminator_loss"], "->", metrics[-1]["discriminator_loss"])


if __name__ == "__main__":
    main()
