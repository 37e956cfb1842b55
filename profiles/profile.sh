#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# usage: bash profiles/profile.sh <tag> [extra bench args]
# Output: gpurun_out/prof_<tag>/ ; summarise with profiles/summarize.py and copy into profiles/.
set -e
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
BARGS="--no-cpu-baseline --no-sac-probe --no-secondary $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 10 --warmup 2 $BARGS > $OUT/bench_kt.json 2> $OUT/kt.err
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 4 --warmup 1 $BARGS > /dev/null 2> $OUT/$name.err; }
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass pmc_sq2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS
pass pmc_sq3 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass pmc_grbm GRBM_GUI_ACTIVE
python3 bench.py --no-sac-probe $@ > $OUT/bench_plain.json 2> $OUT/bench_plain.err
cat $OUT/bench_plain.json
