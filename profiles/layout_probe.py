#!/usr/bin/env python3
"""Does the placement of the rollout's five streams (obs, reward, terminated, truncated, actions) in device
memory change the kernel time?  bench.py runs of the same build differ by several per cent from process to
process on one box; this probe times the bench workload (262144 envs x 250 steps) in ONE process with the
streams carved out of one arena at chosen relative offsets, interleaved over rounds.
    python3 profiles/layout_probe.py [--rounds 4]"""
import ctypes, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd import _capi

n, H = 262144, 250
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 4
cfg = pkg.load_env_config("single_food_long_horizon")
dev = torch.device("cuda", 0)
lib = _capi.load_library()
c = cfg.to_c(); h = ctypes.c_void_p()
_capi.check(lib, lib.salp_vec_create(ctypes.byref(c), n, 0, 0, 0, ctypes.byref(h)), "create")

SZ = dict(obs=H * n * 96, rew=H * n * 4, term=H * n, trunc=H * n, act=H * n * 4)
MiB = 1 << 20
arena = torch.empty(SZ["obs"] + 4 * SZ["rew"] + 512 * MiB, dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)          # 2-MiB aligned start


def layout(skews):
    """Streams back to back from the arena start, each start rounded up to 2 MiB and then moved by its skew."""
    p, out = base + 64 * MiB, {}
    for k in ("obs", "rew", "term", "trunc", "act"):
        p = (p + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
        out[k] = p + skews.get(k, 0)
        p = out[k] + SZ[k]
    assert p < arena.data_ptr() + arena.numel()
    return out


LAYOUTS = {
    "aligned_2MiB": {},
    "skew_4K_steps": dict(rew=4096, term=8192, trunc=12288, act=16384),
    "skew_64K_steps": dict(rew=65536, term=131072, trunc=196608, act=262144),
    "skew_odd_256B": dict(rew=256, term=768, trunc=1280, act=1792),
    "skew_1MiB_thirds": dict(rew=349440, term=699136, trunc=174848, act=524288),
    "obs_shift_512K": dict(obs=524288),
    "obs_shift_1MiB+4K": dict(obs=MiB + 4096, rew=4096 * 3, term=4096 * 5, trunc=4096 * 7, act=4096 * 11),
    "act_far_+96MiB": dict(act=96 * MiB),
}
ptrs = {k: layout(v) for k, v in LAYOUTS.items()}
vp = ctypes.c_void_p
stream = vp(torch.cuda.current_stream().cuda_stream)
# actions: same values in every layout
src = torch.rand((H * n,), device=dev) * 2 - 1


def put_actions(p):      # the layouts overlap inside the arena: refresh this layout's action block before using it
    off = p["act"] - arena.data_ptr()
    arena[off:off + SZ["act"]].view(torch.float32).copy_(src)



def launch(p):
    _capi.check(lib, lib.salp_vec_rollout(h, vp(p["act"]), H, vp(p["obs"]), vp(p["rew"]), vp(p["term"]), vp(p["trunc"]),
                                          None, None, 1, stream), "rollout")


times = {k: [] for k in ptrs}
for k in ptrs:
    put_actions(ptrs[k])
    for _ in range(3):
        launch(ptrs[k])
torch.cuda.synchronize()
for r in range(rounds):
    for k in ptrs:
        put_actions(ptrs[k])
        torch.cuda.synchronize()
        for _ in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); launch(ptrs[k]); e.record(); e.synchronize()
            times[k].append(s.elapsed_time(e))
for k, t in times.items():
    print(json.dumps({"layout": k, "median_ms": round(statistics.median(t), 4), "min_ms": round(min(t), 4), "max_ms": round(max(t), 4),
                      "offsets_mod_2MiB": {s: (ptrs[k][s] % (2 * MiB)) for s in ptrs[k]}}), flush=True)
