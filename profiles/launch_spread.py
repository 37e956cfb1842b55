#!/usr/bin/env python3
"""Per-launch durations of the fused rollout kernel next to what each launch simulated: HIP-event
time of every launch of a bench-like run from reset, the launch's episode / truncation / termination / food counts
(salp_vec_get_stats deltas) and, with a -DSALP_EXP_STAMPS build, the mean per-phase wavefront cycles of that launch.
usage: python profiles/launch_spread.py lib.so [--preset single_food_long_horizon] [--launches 24] [--stamps]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd import _capi

def main():
    path, preset, launches, stamps, n, H = None, "single_food_long_horizon", 24, False, 262144, 250
    it = iter(sys.argv[1:])
    for a in it:
        if a == "--preset": preset = next(it)
        elif a == "--launches": launches = int(next(it))
        elif a == "--stamps": stamps = True
        else: path = os.path.abspath(a)
    cfg = pkg.load_env_config(preset)
    lib = _capi.load_library(path)
    if stamps:
        lib.salp_exp_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.salp_exp_read_stamps.restype = ctypes.c_int
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    act = torch.rand((H, n, cfg.act_dim), generator=gen, device=dev) * 2 - 1       # the bench's action block
    obs = torch.empty((H, n, cfg.obs_dim), device=dev); rew = torch.empty((H, n), device=dev)
    term = torch.empty((H, n), dtype=torch.uint8, device=dev); trunc = torch.empty((H, n), dtype=torch.uint8, device=dev)
    c = cfg.to_c(); h = ctypes.c_void_p()
    _capi.check(lib, lib.salp_vec_create(ctypes.byref(c), n, 0, 0, 0, ctypes.byref(h)), "create")
    vp = ctypes.c_void_p
    def stats():
        s = _capi.CStats(); _capi.check(lib, lib.salp_vec_get_stats(h, ctypes.byref(s)), "stats")
        return {k: getattr(s, k) for k, _ in _capi.CStats._fields_}
    prev = stats(); rows = []
    for i in range(launches):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        _capi.check(lib, lib.salp_vec_rollout(h, vp(act.data_ptr()), H, vp(obs.data_ptr()), vp(rew.data_ptr()), vp(term.data_ptr()),
                    vp(trunc.data_ptr()), None, None, 1, vp(torch.cuda.current_stream().cuda_stream)), "rollout")
        e.record(); e.synchronize()
        cur = stats()
        row = {"launch": i, "steps": f"{i * H}..{(i + 1) * H - 1}", "ms": round(s.elapsed_time(e), 4),
               "episodes": cur["episodes"] - prev["episodes"], "truncated": cur["truncated"] - prev["truncated"],
               "terminated": cur["terminated"] - prev["terminated"], "food": cur["food_collected"] - prev["food_collected"]}
        prev = cur
        if stamps:
            waves = min(n // 64, 8192)
            buf = np.zeros(waves * 16, np.uint32)
            assert lib.salp_exp_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
            raw = buf.reshape(waves, 16)
            ph = raw[:, :12].astype(np.float64).mean(0) / H
            real = ((raw[:, 13].astype(np.int64) - raw[:, 12].astype(np.int64)) & 0xFFFFFFFF) / 100.0
            start = ((raw[:, 12].astype(np.int64) - int(raw[:, 12].min())) & 0xFFFFFFFF) / 100.0
            row["phase_cycles_per_step"] = [round(float(x), 1) for x in ph]
            row["wave_wall_us_p50_p99_max"] = [round(float(np.percentile(real, q)), 1) for q in (50, 99, 100)]
            row["last_start_us"] = round(float(start.max()), 1); row["last_end_us"] = round(float((start + real).max()), 1)
            row["clock_ghz"] = round(float((raw[:, :12].astype(np.float64).sum(1) / np.maximum(real * 100, 1) * 0.1).mean()), 3)
        rows.append(row)
        print(json.dumps(row), flush=True)
    ms = [r["ms"] for r in rows]
    print(json.dumps({"preset": preset, "launches": launches, "min_ms": min(ms), "mean_ms": sum(ms) / len(ms), "max_ms": max(ms),
                      "corr_ms_vs_episodes": float(np.corrcoef(ms, [r["episodes"] for r in rows])[0, 1])}))

if __name__ == "__main__":
    main()
