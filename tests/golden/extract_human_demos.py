#!/usr/bin/env python3
"""Extracts known-answer data from the reference's recorded human demonstrations
(data/expert_demos/human/*.pkl, format in data/expert_demos/README.md:57-75) WITHOUT unpickling:
the files are walked with `pickletools.genops` (a disassembler — nothing from the file is executed,
no object is constructed) and the raw ndarray payloads (BINBYTES after a shape tuple and a dtype
code) are copied out with numpy.frombuffer.

Writes tests/golden/human_demo_<n>.npz with `actions` f64 [T] and `base_obs` f32 [T+1, 10]
(row 0 = observation before the first action, row t+1 = after action t), truncated to the first
T steps.  Columns 4..9 of the base observation (heading, angular velocity, body size, breathing
phase, water volume, nozzle) are functions of the actions alone until the first wall contact, so
they pin the breathing / nozzle / torque arithmetic of the hot path bit-for-bit."""
import glob
import os
import pickletools
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("SALP_REFERENCE_ROOT", "/root/reference") + "/data/expert_demos/human"
KEYS = ("observations", "actions", "rewards", "next_observations", "dones")
T_KEEP = 1500


DTYPES = {"observations": "<f4", "next_observations": "<f4", "actions": "<f8", "rewards": "<f8", "dones": "|b1"}


def arrays_of(path):
    """key -> ndarray.  Per key the stream holds: the key string, the _reconstruct state
    (version, shape ints), a dtype (spelled out the first time, a memo reference later) and one
    BINBYTES payload; the element size is checked against README.md:57-75's dtypes."""
    data = open(path, "rb").read()
    out, key, ints = {}, None, []
    for op, arg, pos in pickletools.genops(data):
        n = op.name
        if n in ("SHORT_BINUNICODE", "BINUNICODE") and arg in KEYS:
            key, ints = arg, []
        elif n in ("BININT", "BININT1", "BININT2") and key:
            ints.append(arg)
        elif n in ("BINBYTES", "BINBYTES8") and key:
            # ints = [0 (ndarray.__new__ shape), 1 (state version), *shape, ...dtype-state ints]
            dt = np.dtype(DTYPES[key])
            count = len(arg) // dt.itemsize
            shape = None
            for cand in ((ints[2], ints[3]) if len(ints) > 3 else None, (ints[2],)):
                if cand and int(np.prod(cand)) == count:
                    shape = cand
                    break
            assert shape is not None and count * dt.itemsize == len(arg), (key, ints[:6], len(arg))
            out[key] = np.frombuffer(arg, dtype=dt).reshape(shape)
            key = None
    return out


def main():
    files = sorted(glob.glob(os.path.join(SRC, "*.pkl")))
    if not files:
        raise SystemExit(f"no demos under {SRC}")
    for i, f in enumerate(files):
        d = arrays_of(f)
        obs, nxt, act = d["observations"], d["next_observations"], d["actions"]
        assert obs.shape[1] == 24 and np.array_equal(nxt[:-1], obs[1:])
        T = min(T_KEEP, len(act))
        base = np.concatenate([obs[:1, :10], nxt[:T, :10]]).astype(np.float32)
        a = np.asarray(act[:T], dtype=np.float64).reshape(T)     # recorded as fp64 (multiples of 0.03 etc.)
        # `observations` / `act32`: the (obs[24], action) pairs the reference's GAIL path samples
        # (training/expert_buffer.py:73-102), first T steps
        np.savez_compressed(os.path.join(HERE, f"human_demo_{i}.npz"), actions=a, base_obs=base,
                            observations=np.ascontiguousarray(obs[:T]).astype(np.float32), act32=a.astype(np.float32).reshape(T, 1),
                            source=np.array(os.path.basename(f)), episode_length=np.array(len(act)))
        print(os.path.basename(f), "steps", len(act), "kept", T, "act dtype", act.dtype, "obs dtype", obs.dtype)


if __name__ == "__main__":
    main()
