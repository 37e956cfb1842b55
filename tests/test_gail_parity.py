"""GAIL parity: `gail.Discriminator` / `gail.ExpertBuffer` against the reference's own classes.

tests/golden/gail_discriminator.npz and gail_expert_buffer.npz were produced by tests/golden/gen_gail_golden.py, which
runs src/salp/agents/discriminator.py (Discriminator.forward / predict_reward / update, :43-139) and
src/salp/training/expert_buffer.py (ExpertBuffer.add_episode / sample, :34-102) in the build container.  Same
weights in, then: forward and predict_reward equal to 1e-6, and after each of three `update` steps on the same batches
the reference's six metrics and — at the end — every parameter, to 1e-6 (float32 arithmetic, different but equivalent
expression of the loss)."""
import os

import numpy as np
import pytest
import torch

from underwater_swimmer_rl_amd.gail import Discriminator, ExpertBuffer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-6


def _check_discriminator(device):
    z = np.load(os.path.join(GOLD, "gail_discriminator.npz"), allow_pickle=False)
    obs_dim, act_dim, B, n_updates, *hidden = [int(v) for v in z["meta"]]
    d = Discriminator(obs_dim, act_dim, hidden, learning_rate=float(z["lr"]), device=device)
    d.load_reference_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w0_")})
    t = lambda k: torch.from_numpy(z[k]).to(device)
    with torch.no_grad():
        prob = d.forward(t("obs"), t("act"))
    rew = d.predict_reward(t("obs"), t("act"))
    assert prob.shape == (B, 1) and rew.shape == (B, 1)                       # the reference's shapes
    assert np.abs(prob.cpu().numpy() - z["prob"]).max() <= TOL
    assert np.abs(rew.cpu().numpy() - z["reward"]).max() <= 5e-6 * max(1.0, float(np.abs(z["reward"]).max()))
    keys = [str(k) for k in z["metric_keys"]]
    expert = {"observations": t("expert_obs"), "actions": t("expert_act")}
    agent = {"observations": z["agent_obs"], "actions": z["agent_act"]}       # numpy batches are accepted, as in the reference
    for i in range(n_updates):
        m = Discriminator.metrics_to_host(d.update(expert, agent))
        assert sorted(m) == keys
        got = np.array([m[k] for k in keys])
        assert np.abs(got - z["metrics"][i]).max() <= 2e-6, (i, dict(zip(keys, got - z["metrics"][i])))
    assert d.training_step == n_updates
    ref_sd = d.reference_state_dict()
    for k in z.files:
        if k.startswith("w3_"):
            w = ref_sd[k[3:]].detach().cpu().numpy()
            assert w.shape == z[k].shape and np.abs(w - z[k]).max() <= TOL, k
            assert np.abs(z[k] - z["w0_" + k[3:]]).max() > 1e-5              # the steps did move the weights


def test_discriminator_matches_the_reference_cpu():
    _check_discriminator("cpu")


@pytest.mark.gpu
def test_discriminator_matches_the_reference_on_the_gpu():
    _check_discriminator("cuda:0")


def test_expert_buffer_sample_contract_of_the_reference():
    z = np.load(os.path.join(GOLD, "gail_expert_buffer.npz"), allow_pickle=False)
    eb = ExpertBuffer(24, 1, device="cpu")
    for i in range(2):
        eb.add_episode({k: z[f"ep{i}_{k}"] for k in ExpertBuffer.KEYS})     # the reference's add_episode(dict) form
    assert len(eb) == eb.num_transitions == int(z["num_transitions"]) and eb.episodes == 2
    s = eb.sample(32, indices=z["indices"])        # the reference's own np.random.randint draw
    assert set(s) == set(ExpertBuffer.KEYS)
    for k in ExpertBuffer.KEYS:
        ref = z["sample_" + k]
        assert tuple(s[k].shape) == ref.shape and np.array_equal(s[k].numpy(), ref.astype(np.float32)), k
    # the library's own draw: uniform over all transitions, the five arrays indexed by the same rows
    torch.manual_seed(0)
    s = eb.sample(4096)
    allo = torch.cat([torch.from_numpy(z[f"ep{i}_observations"]) for i in range(2)])
    alla = torch.cat([torch.from_numpy(z[f"ep{i}_actions"]) for i in range(2)])
    row = (s["observations"][:, None, :] == allo[None]).all(-1).float().argmax(1)
    assert torch.equal(alla[row], s["actions"])
    assert len(torch.unique(row)) == len(eb)                                   # every transition is reachable
    with pytest.raises(ValueError):
        ExpertBuffer(24, 1, device="cpu").sample(4)                            # empty buffer raises, as the reference
