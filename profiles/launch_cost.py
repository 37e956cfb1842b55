#!/usr/bin/env python3
"""Kernel time of one fused rollout launch as a function of its length H (262144 envs, bench preset): T(H) = a + b H.
`a` is what a launch boundary costs (state load / store, ramp-up while 4096 wavefronts start in lockstep)."""
import ctypes, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd import _capi

n, HMAX = 262144, 1000
cfg = pkg.load_env_config("single_food_long_horizon")
dev = torch.device("cuda", 0)
lib = _capi.load_library()
c = cfg.to_c(); h = ctypes.c_void_p()
_capi.check(lib, lib.salp_vec_create(ctypes.byref(c), n, 0, 0, 0, ctypes.byref(h)), "create")
act = torch.rand((HMAX, n, 1), device=dev) * 2 - 1
obs = torch.empty((HMAX, n, 24), device=dev)
rew = torch.empty((HMAX, n), device=dev)
term = torch.empty((HMAX, n), dtype=torch.uint8, device=dev)
trunc = torch.empty((HMAX, n), dtype=torch.uint8, device=dev)
vp = ctypes.c_void_p
st = vp(torch.cuda.current_stream().cuda_stream)


def launch(H):
    _capi.check(lib, lib.salp_vec_rollout(h, vp(act.data_ptr()), H, vp(obs.data_ptr()), vp(rew.data_ptr()), vp(term.data_ptr()),
                                          vp(trunc.data_ptr()), None, None, 1, st), "rollout")


for _ in range(4):
    launch(250)
torch.cuda.synchronize()
Hs = [1, 2, 5, 10, 25, 50, 100, 250, 500, 1000]
times = {H: [] for H in Hs}
for r in range(5):
    for H in Hs:
        for _ in range(3 if H >= 100 else 8):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); launch(H); e.record(); e.synchronize()
            times[H].append(s.elapsed_time(e))
med = {H: statistics.median(t) for H, t in times.items()}
for H in Hs:
    print(json.dumps({"H": H, "median_ms": round(med[H], 4), "us_per_step": round(med[H] / H * 1e3, 3)}), flush=True)
b = (med[1000] - med[250]) / 750
print(json.dumps({"per_step_us_from_250_to_1000": round(b * 1e3, 3), "intercept_us_at_250": round((med[250] - 250 * b) * 1e3, 1),
                  "intercept_us_at_25": round((med[25] - 25 * b) * 1e3, 1)}))
