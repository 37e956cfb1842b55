"""Builds libsalp_hip.so (gfx950) in-tree with hipcc.  `python build.py [--force]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "salp_vec.hip")             # SalpSnakeEnv.step hot path
SRC_ROBOT = os.path.join(HERE, "salp_robot.hip")     # HEAD Robot simulator
DEPS = [SRC, SRC_ROBOT, os.path.join(HERE, "salp_device.h"), os.path.join(HERE, "salp_food_lds.h"),
        os.path.join(HERE, "..", "..", "include", "salp_vec.h"),
        os.path.join(HERE, "..", "..", "include", "salp_robot.h")]
OUT = os.path.join(HERE, "libsalp_hip.so")
# -ffp-contract=off: the fp64 state update must round exactly where the reference rounds
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.isfile(c):
            return c
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False, out: str = OUT, defines=()) -> str:
    """`out`/`defines` build experiment variants (profiles/ab_bench.py); the product is the default."""
    if not force and os.path.isfile(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS):
        return out
    cmd = [hipcc()] + FLAGS + [f"-D{d}" for d in defines] + ["-o", out, SRC, SRC_ROBOT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    out, defs = OUT, []
    for a in args:
        if a.startswith("--out="):
            out = os.path.abspath(a[6:])
        elif a.startswith("-D"):
            defs.append(a[2:])
    print(build(force="--force" in sys.argv, verbose=True, out=out, defines=defs))
