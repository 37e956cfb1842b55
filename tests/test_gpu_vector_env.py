"""The Python VectorEnv shim on the GPU (device-pointer ABI, torch tensors) against the oracle, plus the
SB3 VecEnv adapter and the food-capture / respawn / efficiency-bonus events at batch scale."""
import numpy as np
import pytest

import oracle_lib as ol
import underwater_swimmer_rl_amd as pkg
from golden_util import obs_diff
from underwater_swimmer_rl_amd import _capi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_torch_vector_env_step_and_rollout_match_oracle():
    cfg = pkg.load_env_config("single_food")
    n, seed = 3000, 4
    env = pkg.SalpVectorEnv(cfg, n, device="cuda:0", seed=seed)
    assert env.single_observation_space.shape == (24,) and env.single_action_space.shape == (1,)
    orc = ol.OracleVec(cfg, n, seed=seed)
    obs, info = env.reset()
    assert obs.is_cuda and obs.shape == (n, 24)
    ref0 = orc.reset()
    assert obs_diff(cfg, obs.cpu().numpy(), ref0).max() <= 1e-5
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(60):
        a = torch.rand((n, 1), generator=g, device="cuda") * 2 - 1
        o, r, te, tr, inf = env.step(a)
        ref = orc.step(a.cpu().numpy())
        assert o.dtype == torch.float32 and te.dtype == torch.bool and r.shape == (n,)
        assert obs_diff(cfg, o.cpu().numpy(), ref["obs"]).max() <= 1e-5
        assert np.array_equal(te.cpu().numpy(), ref["terminated"].astype(bool))
        assert np.array_equal(inf["steps_since_food"].cpu().numpy(), ref["info"][:, 1])
    acts = torch.rand((200, n, 1), generator=g, device="cuda") * 2 - 1
    out = env.rollout(acts)
    ref = orc.rollout(acts.cpu().numpy())
    assert obs_diff(cfg, out["obs"].cpu().numpy(), ref["obs"]).max() <= 1e-5
    assert np.array_equal(out["terminated"].cpu().numpy(), ref["terminated"])
    assert env.global_step == 260
    # side stream: the call is asynchronous on the caller's current stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out2 = env.rollout(acts[:20])
    s.synchronize()
    ref2 = orc.rollout(acts[:20].cpu().numpy())
    assert obs_diff(cfg, out2["obs"].cpu().numpy(), ref2["obs"]).max() <= 1e-5
    env.close()


def test_food_capture_respawn_and_bonus_at_scale():
    """Food injected ahead of every swimmer (set_state, as eval/collect_navigation_data.py:76-89):
    thousands of captures, respawn draws (incl. rejected attempts) and efficiency bonuses."""
    cfg = pkg.load_env_config("sac_gail", num_food_items=3, proximity_reward_weight=1.5)
    n, seed, H = 4096, 9, 700
    env = pkg.SalpVectorEnv(cfg, n, device="cuda:0", seed=seed)
    orc = ol.OracleVec(cfg, n, seed=seed)
    f64, i32 = env.get_state()
    rng = np.random.default_rng(1)
    f64[_capi.F_THETA] = rng.uniform(-0.4, 0.4, n)
    f64[_capi.F_FOOD0 + 0] = 400 + rng.uniform(50, 70, n)     # food 0 ahead
    f64[_capi.F_FOOD0 + 3] = 300 + rng.uniform(-10, 10, n)
    f64[_capi.F_FOOD0 + 1] = 400 + rng.uniform(90, 140, n)    # food 1 further ahead
    f64[_capi.F_FOOD0 + 4] = 300 + rng.uniform(-25, 25, n)
    env.set_state(f64, i32)
    fo, io = orc.get_state()
    fo[:] = f64
    orc.set_state(fo, i32)
    act = (rng.uniform(-0.15, 0.15, size=(H, n, 1))).astype(np.float32)
    out = env.rollout(act)
    ref = orc.rollout(act)
    st = env.stats()
    assert st["food_collected"] > 2000, st
    assert np.array_equal(out["terminated"].cpu().numpy(), ref["terminated"])
    assert np.array_equal(out["truncated"].cpu().numpy(), ref["truncated"])
    assert obs_diff(cfg, out["obs"].cpu().numpy(), ref["obs"]).max() <= 1e-5
    r, rr = out["reward"].cpu().numpy().astype(np.float64), ref["reward64"]
    assert (np.abs(r - rr) / np.maximum(1.0, np.abs(rr))).max() <= 1e-5
    assert rr.max() > 1000.0          # food_reward 15 + efficiency_bonus 2 * steps remaining
    fd, idv = env.get_state()
    fo, io = orc.get_state()
    assert np.array_equal(idv, io)    # incl. the per-env draw counters after all the respawns
    assert np.nanmax(np.abs(fd - fo)) <= 1e-9
    assert abs(st["reward_sum"] - float(rr.sum())) <= 1e-6 * abs(rr.sum()) + 1.0
    env.close()


def test_sb3_vecenv_adapter():
    venv = pkg.SalpSB3VecEnv("single_food", num_envs=6, device=0, seed=2, max_steps_without_food=30)
    obs = venv.reset()
    assert obs.shape == (6, 24) and obs.dtype == np.float32
    assert venv.observation_space.shape == (24,) and venv.action_space.shape == (1,)
    saw_done = False
    for t in range(40):
        obs, rew, dones, infos = venv.step(np.zeros((6, 1), np.float32))
        assert obs.shape == (6, 24) and rew.shape == (6,) and dones.dtype == bool and len(infos) == 6
        if dones.any():
            saw_done = True
            i = int(np.nonzero(dones)[0][0])
            assert "terminal_observation" in infos[i] and infos[i]["TimeLimit.truncated"]
            assert infos[i]["terminal_observation"].shape == (24,)
            assert obs[i, 0] == 0.5 and obs[i, 1] == 0.5      # already the next episode's first observation
    assert saw_done
    assert venv.get_attr("max_steps_without_food") == [30] * 6
    assert venv.env_is_wrapped(object) == [False] * 6
    venv.close()


def test_step_is_graph_capturable():
    """The device-pointer entry points do no allocation / synchronisation, so a policy + env step
    loop can be captured in a HIP graph (torch.cuda.CUDAGraph) and replayed."""
    cfg = pkg.load_env_config("single_food")
    n, seed = 2048, 12
    env = pkg.SalpVectorEnv(cfg, n, device="cuda:0", seed=seed)
    orc = ol.OracleVec(cfg, n, seed=seed)
    w = torch.randn(24, 1, device="cuda") * 0.3
    obs, _ = env.reset()
    act = torch.zeros((n, 1), device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):          # warm-up on the side stream (buffers allocated before capture)
        act.copy_(torch.tanh(obs @ w))
        env.step(act, want_final_observation=False)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    orc.reset()                         # the oracle follows the same reset sequence
    env.reset()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        act.copy_(torch.tanh(env._bufs["obs"] @ w))
        env.step(act, want_final_observation=False)
    # the capture itself does not execute; state is still the reset state
    orc.reset()
    for t in range(30):
        g.replay()
        torch.cuda.synchronize()
        a = act.cpu().numpy()
        out = orc.step(a)
        assert obs_diff(cfg, env._bufs["obs"].cpu().numpy(), out["obs"]).max() <= 1e-5, t
    env.close()


def test_sharded_env_on_rccl_world_size_one():
    """The RCCL code path of ShardedSalpVectorEnv on one GPU (world size 1): gathers are identities."""
    import os
    import torch.distributed as dist
    from underwater_swimmer_rl_amd.sharded import ShardedSalpVectorEnv
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = pkg.load_env_config("sac_gail")
        n, seed = 1024, 6
        senv = ShardedSalpVectorEnv(cfg, n, device="cuda:0", seed=seed)
        orc = ol.OracleVec(cfg, n, seed=seed)
        g_obs, _ = senv.reset()
        assert obs_diff(cfg, g_obs.cpu().numpy(), orc.reset()).max() <= 1e-5
        act = torch.rand((40, n, 1), device="cuda") * 2 - 1
        g_obs, g_rew, g_term, g_trunc, _ = senv.step(act[0])
        ref = orc.step(act[0].cpu().numpy())
        assert obs_diff(cfg, g_obs.cpu().numpy(), ref["obs"]).max() <= 1e-5 and g_term.dtype == torch.bool
        out, g_final = senv.rollout(act[1:], gather="final", async_gather=True)
        senv.wait_gather()
        ref = orc.rollout(act[1:].cpu().numpy())
        assert obs_diff(cfg, g_final.cpu().numpy(), ref["obs"][-1]).max() <= 1e-5
        senv.close()
    finally:
        dist.destroy_process_group()


def test_base_num_food_items_poke_like_the_curriculum():
    """continuous_trainer.py:409-411 writes env.base_num_food_items; later resets place that many foods."""
    import torch
    env = pkg.SalpVectorEnv("sac_gail", num_envs=300, device="cuda:0", seed=4, num_food_items=6, max_steps_without_food=30)
    obs, _ = env.reset()
    assert env.base_num_food_items == 6 and float(obs[:, 22].min()) == pytest.approx(0.6)
    assert (env.num_food_items == 6).all()             # what continuous_trainer.py:380 reads, per env
    env.base_num_food_items = 2
    with pytest.raises(Exception):
        env.base_num_food_items = 7                    # more than the slots the env was created with
    act = torch.zeros((300, 1), device="cuda")
    for _ in range(40):                                # every env truncates at step 31 and resets with 2 foods
        obs, *_ = env.step(act)
    assert float(obs[:, 22].max()) <= 0.2 + 1e-6
    assert (env.num_food_items == 2).all()
    sb3 = pkg.SalpSB3VecEnv("sac_gail", num_envs=4, device=0, num_food_items=5)
    sb3.set_attr("base_num_food_items", 3)
    assert sb3.get_attr("base_num_food_items") == [3, 3, 3, 3]
    with pytest.raises(AttributeError):
        sb3.set_attr("food_reward", 1.0)
    env.close(); sb3.close()


def test_c_abi_demo_builds_and_runs_from_plain_c(tmp_path):
    """examples/c_abi_demo.c: the boundary used from C99 with host pointers (no Python, no torch in the process)."""
    import os, shutil, subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "underwater-swimmer_rl_amd", "csrc")
    exe = str(tmp_path / "c_abi_demo")
    subprocess.run([gcc, "-std=c99", "-O2", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_demo.c"),
                    "-o", exe, "-L", libdir, "-lsalp_hip", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    assert "env-steps 1228800" in out, out
