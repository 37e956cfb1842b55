#!/bin/bash
# gfx950 assembly of csrc/salp_vec.hip with the product flags: bash profiles/isa_dump.sh out.s [-DNAME ...]
# Feed to profiles/isa_regs.py / isa_stats.py / isa_ophist.py / isa_cndruns.py.
set -e
OUT=$1; shift
cd "$(dirname "$0")/../underwater-swimmer_rl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
  -mllvm -disable-machine-licm "$@" -S --cuda-device-only -o "$OUT" salp_vec.hip
