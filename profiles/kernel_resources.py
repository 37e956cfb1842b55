#!/usr/bin/env python3
"""Registers / LDS / scratch / resident workgroups per CU of the rollout kernels as the runtime reports them
(salp_vec_last_kernel_resources), per food count, output signature and constants.  On the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import underwater_swimmer_rl_amd as pkg
from underwater_swimmer_rl_amd._capi import SalpLib
for foods in (1, 3, 5, 8, 12, 16):
    for tank in (False, True):
        cfg = pkg.load_env_config("sac_gail", num_food_items=foods, **(dict(width=801) if tank else {}))
        n, H = 2048, 2
        dev = SalpLib(cfg, n, device_id=0, seed=0)
        act = np.zeros((H, n, cfg.act_dim), np.float32)
        obs = np.empty((H, n, cfg.obs_dim), np.float32); rew = np.empty((H, n), np.float32)
        term = np.empty((H, n), np.uint8); trunc = np.empty((H, n), np.uint8); fin = np.empty((H, n, cfg.obs_dim), np.float32)
        for label, f, r in (("main", None, rew), ("extras", fin, rew), ("partial", None, None)):
            dev.rollout(act, H, obs, r, term, trunc, f, None, 0)
            ll, res = dev.last_launch(), dev.last_kernel_resources()
            print(f"foods {foods:2d} slots {ll['food_slots']:2d} {'801-wide' if tank else 'literal '} {label:8s} "
                  f"vgprs {res['vgprs']:3d} lds {res['lds_bytes']:6d} scratch {res['scratch_bytes']:3d} workgroups/CU {res['workgroups_per_cu']}")
        dev.close()
