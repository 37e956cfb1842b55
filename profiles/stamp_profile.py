#!/usr/bin/env python3
"""Phase profile of the rollout kernel from in-kernel shader-clock stamps (experiment build -DSALP_EXP_STAMPS:
`python3 underwater-swimmer_rl_amd/csrc/build.py --force --out=profiles/ab/stamps.so -DSALP_EXP_STAMPS`).
usage: python profiles/stamp_profile.py profiles/ab/stamps.so [--preset sac_gail] [--envs 262144] [--launches 3]
Prints, per phase, the mean over wavefronts of the cycles spent between the stamps (a wavefront's lifetime: issue,
stalls and the time other wavefronts of the SIMD held the pipes), per step, and its share."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd import _capi

PHASES = ["0 loop top, action", "1 nozzle + breathing state machine", "2 jet thrust block", "3 drag, integrate, walls",
          "4 food pass (12 slots)", "5 capture test, selection, reward, counters", "6 reward / flag stores",
          "7 rare events: pass + selection again", "8 observation", "9 tile write, flush, row stores",
          "10 rare events: entry, statistics", "11 rare events: respawn / reset placement"]

def main():
    path, preset, n, H, launches = None, "sac_gail", 262144, 250, 3
    it = iter(sys.argv[1:])
    for a in it:
        if a == "--preset": preset = next(it)
        elif a == "--envs": n = int(next(it))
        elif a == "--launches": launches = int(next(it))
        else: path = os.path.abspath(a)
    cfg = pkg.load_env_config(preset)
    dev = torch.device("cuda", 0)
    act = torch.rand((H, n, cfg.act_dim), device=dev) * 2 - 1
    obs = torch.empty((H, n, cfg.obs_dim), device=dev); rew = torch.empty((H, n), device=dev)
    term = torch.empty((H, n), dtype=torch.uint8, device=dev); trunc = torch.empty((H, n), dtype=torch.uint8, device=dev)
    lib = _capi.load_library(path)
    lib.salp_exp_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.salp_exp_read_stamps.restype = ctypes.c_int
    c = cfg.to_c(); h = ctypes.c_void_p()
    _capi.check(lib, lib.salp_vec_create(ctypes.byref(c), n, 0, 0, 0, ctypes.byref(h)), "create")
    vp = ctypes.c_void_p
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(launches):
        if i == launches - 1: ev[0].record()
        _capi.check(lib, lib.salp_vec_rollout(h, vp(act.data_ptr()), H, vp(obs.data_ptr()), vp(rew.data_ptr()), vp(term.data_ptr()),
                    vp(trunc.data_ptr()), None, None, 1, vp(torch.cuda.current_stream().cuda_stream)), "rollout")
    ev[1].record(); torch.cuda.synchronize()
    waves = min(n // 64, 8192)
    buf = np.zeros(waves * 16, np.uint32)
    assert lib.salp_exp_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
    raw = buf.reshape(waves, 16)
    a = raw[:, :12].astype(np.float64) / H          # cycles per step, per wavefront
    tot = a.sum(1)
    # wall clock of each wavefront's loop (s_memrealtime, 100 MHz): its clock = shader cycles / wall time; when it started
    real = (raw[:, 13].astype(np.int64) - raw[:, 12].astype(np.int64)) & 0xFFFFFFFF
    start = (raw[:, 12].astype(np.int64) - int(raw[:, 12].min())) & 0xFFFFFFFF
    clock_ghz = raw[:, :12].astype(np.float64).sum(1) / np.maximum(real, 1) * 0.1
    timing = {"wave_wall_us_mean": float(real.mean() / 100), "wave_wall_us_p10_p90": [float(np.percentile(real, 10) / 100), float(np.percentile(real, 90) / 100)],
              "clock_ghz_mean": float(clock_ghz.mean()), "clock_ghz_p10_p90": [float(np.percentile(clock_ghz, 10)), float(np.percentile(clock_ghz, 90))],
              "start_us_percentiles_50_75_90_100": [float(np.percentile(start, q) / 100) for q in (50, 75, 90, 100)],
              "end_us_max": float(((start + real).max()) / 100)}
    r1 = start < 0.5 * start.max() if start.max() > 20000 else np.ones_like(start, bool)    # first round: started early
    pct = lambda v: [float(np.percentile(v, q) / 100) for q in (0, 10, 50, 90, 99, 100)] if v.size else []
    timing["round1_waves"] = int(r1.sum()); timing["round1_wall_us_pct_0_10_50_90_99_100"] = pct(real[r1])
    timing["round2_waves"] = int((~r1).sum()); timing["round2_wall_us_pct"] = pct(real[~r1]); timing["round2_start_us_pct"] = pct(start[~r1])
    timing["round2_end_us_pct"] = pct((start + real)[~r1])
    out = {"preset": preset, "envs": n, "horizon": H, "kernel_ms_last_launch": ev[0].elapsed_time(ev[1]),
           "cycles_per_step_mean": float(tot.mean()), "cycles_per_step_p10_p90": [float(np.percentile(tot, 10)), float(np.percentile(tot, 90))],
           "timing": timing,
           "phases": {PHASES[i]: {"cycles_per_step": float(a[:, i].mean()), "share": float(a[:, i].mean() / tot.mean())} for i in range(12)}}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
