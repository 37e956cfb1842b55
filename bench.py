#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched SalpSnakeEnv.step hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload at --gpus 1 (BASELINE.json configs[2], the configuration the metric is quoted on):
N_envs = 262144, single_food_long_horizon parameters, random actions resident in HBM,
the 5000-step rollout run as K fused-rollout launches of `--chunk` steps each (defaults
K = 20 x 250 = 5000 steps).  One bench "step" = one launch = chunk x N_envs env-steps.
All inputs (state, actions) are resident in HBM when the timed region starts; every output
(observations [chunk, N, 24] f32, rewards, terminated, truncated) is written to HBM.

Workload at --gpus G > 1 (launched by torch.distributed.run, one rank per GPU) = BASELINE.json
configs[3]: N_envs = 1 048 576 IN TOTAL, sharded by env index (1048576 / G per GPU: 131072 at G = 8),
same parameters and launches, and every launch all-gathers over RCCL the observations it returned
(`--gather all`: the whole [chunk, N/G, 24] block of every rank, double-buffered so that the collective
of launch k runs beside launch k + 1; this is the exchange the north_star names and it is xGMI-bound by
construction).  The total is fixed while G grows: "scaling": "strong" (over G = 2, 4, 8).
`--gather final` exchanges only the last step's observation of each launch (what a centralised actor
needs to go on), `--gather none` nothing (data-parallel learners); `--weak` keeps 262144 envs per GPU
instead (configs[2] replicated); `--envs` / `--total-envs` override the counts.  `config.workload`
always says which configuration ran.

Prints ONE JSON line on rank 0.  `roofline` prices the fused rollout kernel against HBM
(algorithmic bytes per env-step = act 4 + obs 96 + reward 4 + flags 2 + 2*state/H, SURVEY.md
§8d); `cpu_baseline` times the CPU oracle (oracle/salp_oracle.c) on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
XGMI_LINK_GBS = 153.0  # per xGMI link (task statement: 7 links x ~153 GB/s per GPU, point-to-point)


def algorithmic_bytes_per_env_step(cfg, horizon: int) -> float:
    """A + O + 4 + 2 + 2S/H with S = 48 + 8F, O = 4*(10+4K+2), A = 4*act_dim."""
    S = 48 + 8 * cfg.num_food_items
    O = 4 * cfg.obs_dim
    A = 4 * cfg.act_dim
    return A + O + 4 + 2 + 2.0 * S / horizon


def cpu_baseline(cfg, budget_s: float):
    """Times the CPU oracle on a bounded sample of the same workload (same parameters, random
    actions, all outputs written).  Rank 0, N = 1 only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # the GPU box gives one-GPU jobs a 16-CPU share
    n, H = 16384, 64
    orc = ol.OracleVec(cfg, n, seed=0, threads=cores)
    rng = np.random.default_rng(0)
    act = rng.uniform(-1, 1, size=(H, n, cfg.act_dim)).astype(np.float32)
    orc.rollout(act)  # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.rollout(act)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 10000:
            break
    rate = reps * n * H / el
    # one-core figure on a smaller sample, for the scalar-port number
    orc1 = ol.OracleVec(cfg, 2048, seed=0, threads=1)
    a1 = np.ascontiguousarray(act[:, :2048])
    orc1.rollout(a1)
    t1 = time.perf_counter()
    r1 = 0
    while time.perf_counter() - t1 < min(3.0, budget_s / 3):
        orc1.rollout(a1)
        r1 += 1
    rate1 = r1 * 2048 * H / (time.perf_counter() - t1)
    out = {
        "value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"C oracle (fp64, libm), OpenMP over envs: {n} envs x {H}-step rollouts x {reps} reps "
                  f"({el:.1f} s), same env parameters, all outputs written",
        "one_core_value": rate1,
    }
    # The reference's own pure-Python step cannot be timed here (its files do not travel to the GPU box): its rate
    # is the one measured in the build container by tests/golden/time_reference.py (BASELINE configs[0]).
    rp = os.path.join(ROOT, "profiles", "reference_cpu_rate.json")
    if os.path.isfile(rp):
        try:
            with open(rp) as f:
                rj = json.load(f)
            out["python_reference"] = {
                "value": rj["env_steps_per_s_median"], "unit": "env-steps/s", "cores": rj.get("cores", 1),
                "label": "measured in the build container, not on this host", "workload": rj.get("what"),
                "cpu": rj.get("cpu"), "python": rj.get("python"), "numpy": rj.get("numpy"), "script": rj.get("script"),
            }
        except Exception:   # noqa: BLE001
            pass
    return out


def sac_first_capture(device, world: int, rank: int, envs: int = 4096, max_steps: int = 600):
    """BASELINE configs[4], outside the timed region and bounded to a few seconds: SAC with the agent block of
    configs/sac_gail.yaml on the HIP VectorEnv (sac_gail preset, `envs` envs per GPU, one hipGraph replay per
    vector step; at world > 1 one replica per rank, gradients all-reduced between graph segments) until any env
    collects its first food.  Returns wall-clock seconds and vector steps; never raises (a failure is recorded)."""
    try:
        import underwater_swimmer_rl_amd as pkg
        from underwater_swimmer_rl_amd.sac import SAC, SACConfig, train_sac_graphed
        env = pkg.SalpVectorEnv("sac_gail", num_envs=envs, device=str(device), seed=0, env_index_base=rank * envs)
        cfg = SACConfig.from_preset("sac_gail")
        cfg.learning_starts = 50
        agent = SAC(env.obs_dim, env.act_dim, cfg, device=str(device), seed=0, data_parallel=world > 1,
                    act_low=env.single_action_space.low, act_high=env.single_action_space.high)
        m = train_sac_graphed(env, agent, max_steps, stop_at_first_food=True)
        env.close()
        # the same run once more in this process (new env, new agent, same seeds): what is left when hipBLASLt, the autograd
        # and optimiser kernels and the graph pools exist already — 1.7 of the 2.1 s of a cold run are one-off library
        # and code-object loads (profiles/sac_startup.py)
        warm_s = None
        try:
            env2 = pkg.SalpVectorEnv("sac_gail", num_envs=envs, device=str(device), seed=0, env_index_base=rank * envs)
            agent2 = SAC(env2.obs_dim, env2.act_dim, cfg, device=str(device), seed=0, data_parallel=world > 1,
                         act_low=env2.single_action_space.low, act_high=env2.single_action_space.high)
            m2 = train_sac_graphed(env2, agent2, max_steps, stop_at_first_food=True)
            env2.close()
            warm_s = m2["first_food_wall_s"] if m2["first_food_vector_step"] == m["first_food_vector_step"] else None
        except Exception:   # noqa: BLE001
            warm_s = None
        return {"seconds": m["first_food_wall_s"], "seconds_second_run_same_process": warm_s,
                "vector_steps": m["first_food_vector_step"], "envs_per_gpu": envs,
                "n_gpus": world, "mode": "hipgraph, segmented + RCCL gradient all-reduce" if world > 1 else "hipgraph",
                "updates": m["updates"], "learn_ms_per_vector_step": m["learn_ms_per_vector_step"],
                "note": "includes graph capture and 3 eager warm-up iterations per phase; outside the timed region"}
    except Exception as e:   # noqa: BLE001 — the bench line must not be lost to the probe
        return {"error": f"{type(e).__name__}: {e}"}


def secondary_kernels(device, envs: int, chunk: int, launches: int = 10, warmup: int = 3):
    """The rollout kernels the headline does not run, timed the same way (HIP events on the launch stream around each
    launch, inputs resident in HBM), AFTER the headline and outside its timed region: the 12-food kernel of the preset
    BASELINE configs[4] names (configs/sac_gail.yaml:9), the same parameters with 16 foods (the 16-slot kernel: the K = 3
    kernel furthest below its roofline) and with 5 (the reference class's default count: the 8-slot kernel).  At most
    `warmup + launches` <= 15 launches per entry."""
    import torch
    import underwater_swimmer_rl_amd as pkg
    from underwater_swimmer_rl_amd.vector_env import SalpVectorEnv
    out = []
    for preset, over in (("sac_gail", {}), ("sac_gail", {"num_food_items": 16}), ("sac_gail", {"num_food_items": 5})):
        try:
            cfg = pkg.load_env_config(preset, **over)
            env = SalpVectorEnv(cfg, envs, device=str(device), seed=0, env_index_base=0)
            gen = torch.Generator(device=device)
            gen.manual_seed(4321)
            act = torch.rand((chunk, envs, cfg.act_dim), generator=gen, device=device, dtype=torch.float32) * 2.0 - 1.0
            for _ in range(warmup):
                env.rollout(act)
            ms = []
            for _ in range(launches):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); env.rollout(act); e.record(); e.synchronize()
                ms.append(s.elapsed_time(e))
            bpe = algorithmic_bytes_per_env_step(cfg, chunk)
            avg = sum(ms) / len(ms)
            ach = bpe * envs * chunk / (avg * 1e-3) / 1e9
            slots = env._lib.last_launch()["food_slots"]
            out.append({"preset": preset, "overrides": over, "kernel": f"salp_rollout_kernel<{slots},3> "
                        f"({cfg.num_food_items} foods in VGPRs + fp32 mirror in LDS)", "envs": envs, "chunk": chunk,
                        "launches": launches, "warmup": warmup, "avg_kernel_ms": avg, "min_kernel_ms": min(ms), "max_kernel_ms": max(ms),
                        "achieved": ach, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS, "bound": "hbm (priced); VALU-issue (actual)",
                        "algorithmic_bytes_per_env_step": bpe, "env_steps_per_launch": envs * chunk,
                        "env_steps_per_s": envs * chunk / (avg * 1e-3), "food_collected": env.stats()["food_collected"]})
            env.close()
            del act
            torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001 — an extra, never the headline
            out.append({"preset": preset, "overrides": over, "error": f"{type(e).__name__}: {e}"})
    return out


def step_mode(device, envs: int, preset: str, iters: int = 200, warmup: int = 20):
    """`salp_vec_step` (one launch per env step, the reference's own call shape) on device tensors, back to back on one
    stream: device time per step by HIP events over `iters` calls.  Accounting: state both ways
    (48 + 8 F bytes each) + action + observation + reward + two flags per env-step."""
    import torch
    from underwater_swimmer_rl_amd.vector_env import SalpVectorEnv
    try:
        env = SalpVectorEnv(preset, num_envs=envs, device=str(device), seed=0)
        act = torch.rand((envs, env.act_dim), device=device) * 2 - 1
        for _ in range(warmup):
            env.step(act, want_final_observation=False)
        torch.cuda.synchronize(device)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            env.step(act, want_final_observation=False)
        e.record(); e.synchronize()
        us = s.elapsed_time(e) / iters * 1e3
        cfg = env.cfg
        bpe = 2 * (48 + 8 * cfg.num_food_items) + 4 * cfg.act_dim + 4 * cfg.obs_dim + 4 + 2
        env.close()
        return {"preset": preset, "envs": envs, "calls": iters, "us_per_step": us, "env_steps_per_s": envs / (us * 1e-6),
                "algorithmic_bytes_per_env_step": bpe, "achieved": bpe * envs / (us * 1e-6) / 1e9, "unit": "GB/s",
                "frac": bpe * envs / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    except Exception as e:   # noqa: BLE001 — an extra, never the headline
        return {"preset": preset, "envs": envs, "error": f"{type(e).__name__}: {e}"}


CONFIG2_ENVS = 262144        # BASELINE configs[2]: one GPU
CONFIG3_TOTAL_ENVS = 1048576  # BASELINE configs[3]: sharded over the GPUs of one node


def shard_plan(world: int, envs_per_gpu=None, total_envs=None, weak: bool = False) -> dict:
    """Which BASELINE configuration a run of `world` ranks measures, and its split by env index.
    Returns envs_per_gpu, total_envs, config (index into BASELINE.json configs, None for an override),
    scaling and env_index_base(rank)."""
    if envs_per_gpu is not None and total_envs is not None:
        raise ValueError("give --envs (per GPU) or --total-envs, not both")
    if envs_per_gpu is not None:
        n, cfg_i, scaling = int(envs_per_gpu), (2 if int(envs_per_gpu) == CONFIG2_ENVS else None), "weak"
    elif total_envs is not None:
        if total_envs % world:
            raise ValueError(f"--total-envs {total_envs} is not divisible by {world} ranks")
        n, cfg_i, scaling = total_envs // world, (3 if total_envs == CONFIG3_TOTAL_ENVS and world > 1 else None), "strong"
    elif world == 1 or weak:
        n, cfg_i, scaling = CONFIG2_ENVS, 2, "weak"
    else:
        if CONFIG3_TOTAL_ENVS % world:
            raise ValueError(f"{CONFIG3_TOTAL_ENVS} envs do not split evenly over {world} ranks; pass --total-envs")
        n, cfg_i, scaling = CONFIG3_TOTAL_ENVS // world, 3, "strong"
    return {"envs_per_gpu": n, "total_envs": n * world, "config": cfg_i, "scaling": scaling,
            "env_index_base": lambda rank: rank * n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: 262144 at --gpus 1, 1048576 / G at --gpus G)")
    ap.add_argument("--total-envs", type=int, default=None, help="envs in total, split evenly over the GPUs")
    ap.add_argument("--weak", action="store_true", help="--gpus G > 1: 262144 envs per GPU (configs[2] replicated) instead of configs[3]")
    ap.add_argument("--chunk", type=int, default=250, help="env-steps per fused rollout launch")
    ap.add_argument("--preset", default="single_food_long_horizon")
    ap.add_argument("--gather", default=None, choices=["final", "all", "none"],
                    help="--gpus G > 1: what each launch exchanges (default all = every returned observation)")
    ap.add_argument("--actions", default="hbm", choices=["hbm", "generated"],
                    help="hbm: a random action block resident in HBM, read by the kernel; generated: the kernel draws the "
                         "actions from each env's Philox action stream and writes them out (same 4 B per env-step)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sac-probe", action="store_true", help="skip the configs[4] first-food-capture probe")
    ap.add_argument("--alt-modes", action="store_true", help="--gpus G > 1: also time --gather final / none after the headline "
                    "(opt-in: the extra collectives have never run on more than one GPU; a failure is agreed across the ranks)")
    ap.add_argument("--no-alt-modes", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary_kernels record (the 12-food kernel, after the headline)")
    ap.add_argument("--probe-seconds", type=float, default=90.0, help="wall-clock limit of the extras that run before the line is printed")
    ap.add_argument("--sac-probe", action="store_true", help="--gpus G > 1: run the configs[4] probe data-parallel on every rank "
                    "(off by default there: that path — graph segments + RCCL gradient all-reduces — is covered by gloo and "
                    "world-size-1 tests only, and a scaling run must not depend on it)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import underwater_swimmer_rl_amd as pkg
    from underwater_swimmer_rl_amd.vector_env import SalpVectorEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")
    # SALP_BENCH_REHEARSAL=1 (one-GPU boxes only): ranks share the visible GPUs (local_rank modulo their number) and
    # talk over gloo, so that the multi-process code path — sharding by rank, barriers, the max-over-ranks timing,
    # rank-0 reporting — can be run where RCCL cannot put two ranks on one device.  Never used by a real run.
    rehearsal = os.environ.get("SALP_BENCH_REHEARSAL") == "1" and world > 1
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # SALP_BENCH_FORCE_SHARDED=1 runs the sharded (RCCL) code path at world size 1 — a rehearsal of the
    # multi-GPU path on a one-GPU box; the reported numbers are then those of that path.
    force_sharded = world == 1 and os.environ.get("SALP_BENCH_FORCE_SHARDED") == "1"
    if world > 1 or force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")   # RCCL's stream ahead of the rollout launches in the dispatcher
        if force_sharded:
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
        elif rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    cfg = pkg.load_env_config(args.preset)
    plan = shard_plan(world, args.envs, args.total_envs, args.weak)
    if args.preset != "single_food_long_horizon" or args.chunk != 250:
        plan["config"] = None        # BASELINE configs[2] / [3] are quoted on this preset in 250-step launches: anything else is an override
    n, H, K, W = plan["envs_per_gpu"], args.chunk, args.steps, args.warmup
    gather = args.gather or "all"

    if world > 1 or force_sharded:
        from underwater_swimmer_rl_amd.sharded import ShardedSalpVectorEnv
        senv = ShardedSalpVectorEnv(cfg, n * world, device=f"cuda:{dev_index}", seed=0)
        env = senv.engine
    else:
        senv = None
        env = SalpVectorEnv(cfg, n, device=f"cuda:{dev_index}", seed=0, env_index_base=0)

    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    act = torch.rand((H, n, cfg.act_dim), generator=gen, device=device, dtype=torch.float32) * 2.0 - 1.0
    if args.actions == "generated":
        act = None      # salp_vec_rollout(act = NULL, act_out = buffer): SURVEY.md §8d config 3 "generated on device"
    rkw = {} if act is not None else {"horizon": H}

    # --gather all: two output blocks, so that the collective of launch k (which reads block k % 2) runs beside
    # launch k + 1 (which writes the other one); a block is only rewritten after its collective has completed
    outs, works = [None, None], [None, None]
    gather_steps = max(1, min(H, int((2 << 30) // max(1, n * cfg.obs_dim * 4))))   # steps per all-gather piece (<= ~2 GB per rank)
    if senv is not None and gather == "all":
        assert senv.env_index_base == plan["env_index_base"](rank) and senv.local_envs == n
        for b in range(2):
            outs[b] = dict(obs=torch.empty((H, n, cfg.obs_dim), device=device), reward=torch.empty((H, n), device=device),
                           terminated=torch.empty((H, n), dtype=torch.uint8, device=device),
                           truncated=torch.empty((H, n), dtype=torch.uint8, device=device))

    def sharded_launch(k, gather=gather):
        if gather == "all":
            b = k & 1
            if works[b] is not None:
                for w_ in works[b]:
                    w_.wait()
                works[b] = None
            st = torch.cuda.Event(enable_timing=True)
            st.record()                      # after the wait for this block's previous collective: the kernel alone
            out = senv.engine.rollout(act, out=outs[b], **rkw)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            # the block goes out in pieces of at most ~2 GB per rank (whole steps): one collective of 12.6 GB per rank
            # (G = 2) is a message size RCCL is rarely run at; gathered layout [piece][G][steps, N/G, obs_dim]
            works[b] = []
            for i, h0 in enumerate(range(0, H, gather_steps)):
                piece = out["obs"][h0:h0 + gather_steps]
                _, w_ = senv.all_gather(f"all_obs{b}_{i}", piece.reshape(1, piece.shape[0], n, cfg.obs_dim), async_op=True)
                works[b].append(w_)
            return st, ev
        st = torch.cuda.Event(enable_timing=True)
        st.record()
        out = senv.engine.rollout(act, **rkw)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if gather == "final":   # staged + double-buffered: the next launch starts at once, the collective runs beside it
            senv.gather_final_async(out["obs"][-1])
        return st, ev

    def drain():
        for b in range(2):
            if works[b] is not None:
                for w_ in works[b]:
                    w_.wait()
                works[b] = None
        if senv is not None:
            senv.wait_gather()

    for k in range(W):
        if senv is not None:
            sharded_launch(k)
        else:
            env.rollout(act, **rkw)
    drain()
    env.clear_stats()

    starts = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    ends = [None] * K
    torch.cuda.synchronize(device)
    if senv is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(K):
        if senv is not None:
            starts[k], ends[k] = sharded_launch(k)
        else:
            starts[k].record()
            env.rollout(act, **rkw)
            ends[k] = torch.cuda.Event(enable_timing=True)
            ends[k].record()
    drain()
    torch.cuda.synchronize(device)
    if senv is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0

    el_t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if senv is not None:
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
    elapsed = float(el_t.item())
    kernel_ms = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3

    stats = env.stats()
    assert stats["env_steps"] == n * H * K, (stats["env_steps"], n * H * K)  # nothing skipped in the timed region

    total_env_steps = float(world) * n * H * K
    value = total_env_steps / elapsed
    bpe = algorithmic_bytes_per_env_step(cfg, H)
    achieved_gbs = bpe * n * H / avg_kernel_s / 1e9

    # HBM bytes per launch from the PMC counters of the committed profile of this workload: WRITE_SIZE + 2 x FETCH_SIZE
    # (on gfx950 FETCH_SIZE reports half the bytes read, MI355X_MICROARCH.md §HBM: the raw figure is BELOW the action
    # bytes the kernel must read); the raw sum is kept beside it.
    traffic = traffic_raw = None
    tp = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.isfile(tp):
        try:
            with open(tp) as f:
                tj = json.load(f)
            if tj.get("envs") == n and tj.get("chunk") == H and tj.get("preset") == args.preset:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_raw = tj.get("hbm_bytes_per_launch_raw")
        except Exception:
            traffic = traffic_raw = None

    line = {
        "metric": "env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
        "scaling": plan["scaling"], "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": (f"BASELINE configs[{plan['config']}]" if plan["config"] is not None else "override (not a BASELINE config)") +
                        f": N_envs={plan['total_envs']} total = {n}/GPU x {world}, {args.preset}, {H * K}-step rollout as {K} fused "
                        f"launches of {H} steps, " +
                        ("random actions in HBM" if act is not None else "random actions drawn in the kernel and written out") +
                        ("" if senv is None else {"all": ", every launch all-gathers ALL observations it returned ([chunk, N/G, obs_dim] per rank) over RCCL",
                                                  "final": ", every launch all-gathers the LAST step's observation over RCCL",
                                                  "none": ", no exchange"}[gather]),
            "baseline_config": plan["config"], "total_envs": plan["total_envs"],
            "envs_per_gpu": n, "chunk": H, "obs_dim": cfg.obs_dim, "act_dim": cfg.act_dim,
            "gather": gather if senv is not None else None,
            "parallelism": f"env-sharded x{world}" + (f", all-gather {gather} obs" if senv is not None else ""),
            "env_steps_per_bench_step": n * H * world,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_raw_fetch": traffic_raw,
            "kernel": "salp_rollout_kernel", "avg_kernel_ms": avg_kernel_s * 1e3,
            "min_kernel_ms": min(kernel_ms), "max_kernel_ms": max(kernel_ms),
            "algorithmic_bytes_per_env_step": bpe, "env_steps_per_launch": n * H,
        },
        "episodes_finished": stats["episodes"], "food_collected": stats["food_collected"],
    }
    if senv is not None:
        # what the launches cost with and without the exchange, and the exchange against the xGMI links it can use
        block = H * n * cfg.obs_dim * 4
        recv = {"all": (world - 1) * block, "final": (world - 1) * n * cfg.obs_dim * 4, "none": 0}[gather]
        line["kernel_side_value"] = float(world) * n * H / avg_kernel_s        # env-steps/s if only the rollout kernels counted
        line["exchange"] = {
            "mode": gather, "recv_bytes_per_rank_per_launch": recv, "wall_s_per_launch": elapsed / K,
            "recv_GBps_per_rank": recv / (elapsed / K) / 1e9 if recv else 0.0,
            "xgmi_peak_GBps_per_rank": XGMI_LINK_GBS * min(max(world - 1, 0), 7),
            "note": "the all-gather of every returned observation moves (G-1) x chunk x N/G x obs_dim x 4 B into each rank per "
                    "launch: the whole-job value is the exchange's rate when that exceeds the kernel time (DESIGN.md §7)",
        }
    if senv is not None and gather == "all" and world > 1 and args.alt_modes:
        # The same launches with the two lighter exchanges, outside the headline's timed region (K // 2 launches each,
        # same barrier / max-over-ranks protocol): what the sharded simulator does when not every observation has
        # to cross xGMI.  Reported beside `value`, never instead of it.  Opt-in.  No new multi-GB allocation (the
        # rollouts write into the headline's output blocks), and every phase ends with an all-reduced error flag, so
        # that a rank that failed and the ranks that did not leave together instead of one of them blocking in a
        # barrier the other never reaches.
        alt = {}
        failed = False
        for mode in ("final", "none"):
            if failed:
                break
            err = None
            e2 = torch.zeros(1, dtype=torch.float64, device=device)
            K2 = max(3, K // 2)
            try:
                def alt_launch(k):
                    out = senv.engine.rollout(act, out=outs[k & 1], **rkw)
                    if mode == "final":
                        senv.gather_final_async(out["obs"][-1])
                for k in range(2):
                    alt_launch(k)
                drain()
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
                for k in range(K2):
                    alt_launch(k)
                drain()
                torch.cuda.synchronize(device)
                e2[0] = time.perf_counter() - t1
            except Exception as e:   # noqa: BLE001 — local failure: reported to every rank by the flag below
                err = f"{type(e).__name__}: {e}"
            flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float64, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)          # reached by every rank, failed or not
            if float(flag.item()) > 0:
                alt[mode] = {"error": err or "another rank failed"}
                failed = True
                continue
            dist.all_reduce(e2, op=dist.ReduceOp.MAX)
            alt[mode] = {"value": float(world) * n * H * K2 / float(e2.item()), "steps": K2,
                         "ms_per_step": float(e2.item()) / K2 * 1e3,
                         "note": "wall time per rank from launch to drained exchange, max over ranks (no barrier inside)"}
        line["other_exchange_modes"] = alt

    # Extras that run before the one JSON line is printed are bounded by a wall-clock limit: if they hang (a GPU fault
    # in the probe, a collective that never completes) the watchdog prints the headline as it stands and ends the process.
    emitted = {"done": False}

    def emit_and_exit():
        if rank == 0 and not emitted["done"]:
            emitted["done"] = True
            line.setdefault("extras_timed_out", True)
            print(json.dumps(line), flush=True)
        os._exit(0)

    import threading
    watchdog = threading.Timer(max(5.0, args.probe_seconds), emit_and_exit)
    watchdog.daemon = True
    watchdog.start()
    if world == 1 and not args.no_secondary and not force_sharded and plan["config"] == 2:
        if senv is None:       # give the headline's 6.7-GB output block back first
            env.close()
            env = None
            torch.cuda.empty_cache()
        line["secondary_kernels"] = secondary_kernels(device, n, H)
        line["step_per_launch"] = [step_mode(device, n, args.preset), step_mode(device, n, "sac_gail"), step_mode(device, 4096, "sac_gail")]
    if not args.no_sac_probe and not rehearsal and not force_sharded and (world == 1 or args.sac_probe):
        probe = sac_first_capture(device, world if senv is not None else 1, rank)   # every rank takes part (collectives)
        line["sac_first_capture"] = probe
    watchdog.cancel()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0 and not emitted["done"]:
        emitted["done"] = True
        print(json.dumps(line), flush=True)
    if senv is not None:
        senv.close()
        dist.destroy_process_group()
    elif env is not None:
        env.close()


if __name__ == "__main__":
    main()
