"""Builds libsalp_hip.so (gfx950) in-tree with hipcc.  `python build.py [--force]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "salp_vec.hip")             # SalpSnakeEnv.step hot path
SRC_ROBOT = os.path.join(HERE, "salp_robot.hip")     # HEAD Robot simulator
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "..", "include"))


def deps() -> list:
    """Every file the library is compiled from: all of csrc/*.h, csrc/*.hip and include/*.h (globbed, so that a new
    header cannot be forgotten — round 2 shipped a list that missed salp_food_reg.h)."""
    import glob
    return sorted(glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "*.hip")) +
                  glob.glob(os.path.join(INCLUDE, "*.h")))


OUT = os.path.join(HERE, "libsalp_hip.so")
# -ffp-contract=off: the fp64 state update must round exactly where the reference rounds
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# Per-source flags.  salp_vec.hip: machine LICM hoists every fp64 literal of the step loop (each one an
# s_mov pair or a v_mov pair) into registers that then live across the whole loop — ~45 VGPRs of polynomial
# coefficients and >100 SGPRs, the latter spilled to VGPR lanes and read back with v_readlane in the loop.
# Without it the one-food kernel needs 92 VGPRs / 76 SGPRs and no scratch (was 128 / 106 + 38 spilled + 20 B
# scratch), the 12-food kernel 107 VGPRs (was 160).  Measured: profiles/r02/ab_notes.md.
SRC_FLAGS = {SRC: ["-mllvm", "-disable-machine-licm"], SRC_ROBOT: []}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.isfile(c):
            return c
    raise RuntimeError("hipcc not found")


def source_hash(defines=(), extra_flags=()) -> str:
    """sha256 over the contents of every dependency, the compiler flags and the defines: what the built library is a
    function of.  Stored next to the library (`<out>.hash`); a library whose stored hash differs is stale whatever the
    file times say (prebuilt .so files travel to the GPU box with the snapshot, where mtimes mean nothing)."""
    import hashlib
    h = hashlib.sha256()
    for d in deps():
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(repr((FLAGS, sorted((os.path.basename(k), v) for k, v in SRC_FLAGS.items()), list(defines), list(extra_flags))).encode())
    return h.hexdigest()


def is_current(out: str = OUT, defines=(), extra_flags=()) -> bool:
    try:
        with open(out + ".hash") as f:
            return os.path.isfile(out) and f.read().strip() == source_hash(defines, extra_flags)
    except OSError:
        return False


def build(force: bool = False, verbose: bool = False, out: str = OUT, defines=(), extra_flags=()) -> str:
    """`out`/`defines`/`extra_flags` build experiment variants (profiles/ab_bench.py); the product is the default."""
    if not force and is_current(out, defines, extra_flags):
        return out
    digest = source_hash(defines, extra_flags)
    objs, procs = [], []
    for src in (SRC, SRC_ROBOT):      # one compile per source (different flags), in parallel, then link
        obj = out + "." + os.path.splitext(os.path.basename(src))[0] + ".o"
        cmd = [hipcc()] + FLAGS + SRC_FLAGS[src] + list(extra_flags) + [f"-D{d}" for d in defines] + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, cwd=HERE)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=HERE)
    for o in objs:
        os.remove(o)
    with open(out + ".hash", "w") as f:
        f.write(digest + "\n")
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    out, defs, extra = OUT, [], []
    for a in args:
        if a.startswith("--out="):
            out = os.path.abspath(a[6:])
        elif a.startswith("-D"):
            defs.append(a[2:])
        elif a.startswith("--flag="):       # e.g. --flag=-mllvm --flag=-disable-machine-licm
            extra.append(a[7:])
    print(build(force="--force" in sys.argv, verbose=True, out=out, defines=defs, extra_flags=extra))
