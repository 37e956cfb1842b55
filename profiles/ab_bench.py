#!/usr/bin/env python3
"""Interleaved A/B timing of several builds of libsalp_hip.so in ONE process (same device, same
data), as cdna_hip_programming.md §5.4 rule 24 asks.  usage:
    python profiles/ab_bench.py name1=path1.so name2=path2.so ... [--rounds 6] [--launches 5]
Prints per-variant median / min kernel ms (HIP events) for the bench workload."""
import ctypes, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd import _capi

def main():
    variants, rounds, launches, n, H, preset = [], 6, 5, 262144, 250, "single_food_long_horizon"
    want_fin = False
    overrides = {}
    it = iter(sys.argv[1:])
    for a in it:
        if a == "--rounds": rounds = int(next(it))
        elif a == "--launches": launches = int(next(it))
        elif a == "--envs": n = int(next(it))
        elif a == "--chunk": H = int(next(it))
        elif a == "--preset": preset = next(it)
        elif a == "--set":                           # --set width=801 (a SalpSnakeEnv keyword on top of the preset)
            k, v = next(it).split("=", 1); overrides[k] = (int(v) if v.lstrip("-").isdigit() else (v == "true") if v in ("true", "false") else float(v))
        elif a == "--final-obs": want_fin = True      # the non-FULL output signature (terminal observations written)
        else:
            k, v = a.split("=", 1); variants.append((k, os.path.abspath(v)))
    # a variant name ending in "+gen" runs the device-generated-action mode (act = NULL, actions
    # written to act_out); "+gennoout" the same without act_out
    cfg = pkg.load_env_config(preset, **overrides)
    dev = torch.device("cuda", 0)
    act = torch.rand((H, n, cfg.act_dim), device=dev) * 2 - 1
    obs = torch.empty((H, n, cfg.obs_dim), device=dev)
    rew = torch.empty((H, n), device=dev)
    term = torch.empty((H, n), dtype=torch.uint8, device=dev)
    trunc = torch.empty((H, n), dtype=torch.uint8, device=dev)
    fin = torch.empty((H, n, cfg.obs_dim), device=dev) if want_fin else None
    handles = {}
    for name, path in variants:
        lib = _capi.load_library(path)
        c = cfg.to_c(); h = ctypes.c_void_p()
        _capi.check(lib, lib.salp_vec_create(ctypes.byref(c), n, 0, 0, 0, ctypes.byref(h)), "create")
        handles[name] = (lib, h, c)
    def launch(name):
        lib, h, _ = handles[name]
        vp = ctypes.c_void_p
        a_in = None if "+gen" in name else vp(act.data_ptr())
        a_out = vp(act.data_ptr()) if name.endswith("+gen") else None
        _capi.check(lib, lib.salp_vec_rollout(h, a_in, H, vp(obs.data_ptr()), vp(rew.data_ptr()),
                    vp(term.data_ptr()), vp(trunc.data_ptr()), vp(fin.data_ptr()) if fin is not None else None, a_out, 1, vp(torch.cuda.current_stream().cuda_stream)), "rollout")
    times = {name: [] for name, _ in variants}
    warm = int(os.environ.get("AB_WARM", "8"))   # advance every variant to the same (desynchronised) phase mix
    for name, _ in variants:
        for _ in range(warm): launch(name)
    torch.cuda.synchronize()
    for r in range(rounds):
        for name, _ in variants:
            for _ in range(launches):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); launch(name); e.record(); e.synchronize()
                times[name].append(s.elapsed_time(e))
    # every variant simulates the same trajectory (same seed, warm-up and launch count), so launch i of one variant and
    # launch i of another do the same work: the MEAN over all launches, and the mean of the per-launch ratios to the
    # first variant, compare like with like even though launches differ from each other (event rates drift)
    base = times[variants[0][0]]
    out = {name: {"median_ms": statistics.median(t), "mean_ms": statistics.fmean(t), "min_ms": min(t), "max_ms": max(t), "n": len(t),
                  "paired_ratio_to_first": statistics.fmean(a / b for a, b in zip(t, base))} for name, t in times.items()}
    if os.environ.get("AB_DUMP") == "1":   # every launch time in order (e.g. with AB_WARM=0: the phase mix desynchronising)
        for name, t in times.items():
            out[name]["all_ms"] = [round(x, 4) for x in t]
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
