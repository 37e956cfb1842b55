#!/bin/bash
# Where does a wavefront of the rollout kernel spend its cycles?  SQ counters of one variant library at a
# given env count (65536 envs = one wavefront per SIMD, 262144 = the bench).  Runs on the GPU box.
# usage: bash profiles/pmc_wave.sh <tag> <lib.so> <preset> <envs>
set -e
TAG=$1; LIB=$2; PRESET=$3; ENVS=$4
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmcw_$TAG
mkdir -p $OUT
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 profiles/ab_bench.py x=$LIB --preset $PRESET --envs $ENVS --rounds 1 --launches 3 > /dev/null 2> $OUT/$name.err; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU
pass b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_IFETCH
pass c SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM
pass d GRBM_GUI_ACTIVE
python3 profiles/summarize.py $OUT $OUT/summary > /dev/null
python3 - <<PY
import json
d=json.load(open("$OUT/summary_summary.json"))
p={k:v['mean_per_launch'] for k,v in d['pmc'].items()}
w=p.get('SQ_WAVES',1)
print("$TAG", "dispatch", d.get('dispatch'))
for k in sorted(p): print(f"  {k:28s} {p[k]:16.0f}  per-wave {p[k]/w:12.1f}")
PY
