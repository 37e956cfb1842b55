#!/usr/bin/env python3
"""Event counters of an experiment build (-DSALP_EXP_COUNT): how often the wave-uniform rare branches of the 12-food
kernel run.  usage: python profiles/count_events.py lib.so [preset] [launches]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd import _capi
path = os.path.abspath(sys.argv[1]); preset = sys.argv[2] if len(sys.argv) > 2 else "sac_gail"
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n, H = 262144, 250
cfg = pkg.load_env_config(preset)
lib = _capi.load_library(path)
c = cfg.to_c(); h = ctypes.c_void_p()
_capi.check(lib, lib.salp_vec_create(ctypes.byref(c), n, 0, 0, 0, ctypes.byref(h)), "create")
dev = torch.device("cuda", 0)
act = torch.rand((H, n, cfg.act_dim), device=dev) * 2 - 1
obs = torch.empty((H, n, cfg.obs_dim), device=dev); rew = torch.empty((H, n), device=dev)
term = torch.empty((H, n), dtype=torch.uint8, device=dev); trunc = torch.empty((H, n), dtype=torch.uint8, device=dev)
vp = ctypes.c_void_p
prev = (ctypes.c_ulonglong * 8)()
lib.salp_exp_read_counters(prev)
prev = list(prev)
for i in range(launches):
    _capi.check(lib, lib.salp_vec_rollout(h, vp(act.data_ptr()), H, vp(obs.data_ptr()), vp(rew.data_ptr()), vp(term.data_ptr()),
                                          vp(trunc.data_ptr()), None, None, 1, vp(torch.cuda.current_stream().cuda_stream)), "rollout")
    torch.cuda.synchronize()
    cur = (ctypes.c_ulonglong * 8)(); lib.salp_exp_read_counters(cur); cur = list(cur)
    d = [a - b for a, b in zip(cur, prev)]; prev = cur
    print(f"launch {i}: wave-steps {d[2]}  tie fallbacks {d[0]} ({d[0] / max(d[2], 1):.4%})  careful {d[1]}  all-live {d[3] / max(d[2], 1):.3%}")
