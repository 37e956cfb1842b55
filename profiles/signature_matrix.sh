#!/bin/bash
# On the GPU box: 250-step launches of 262144 envs for every (foods, output signature, constants) form of the rollout kernel,
# one line each: bash profiles/signature_matrix.sh [lib.so] > gpurun_out/r03/signature_matrix.txt
LIB=${1:-underwater-swimmer_rl_amd/csrc/libsalp_hip.so}
mkdir -p gpurun_out/r03
for F in 1 3 5 8 12 16; do
  for SIG in "" "--final-obs"; do
    for TANK in "" "--set width=801"; do
      python profiles/ab_bench.py cur=$LIB --preset sac_gail --set num_food_items=$F $SIG $TANK --rounds 2 --launches 5 > gpurun_out/r03/_m.json 2> gpurun_out/r03/_m.err || { tail -3 gpurun_out/r03/_m.err; exit 1; }
      python -c "
import json; d=json.load(open('gpurun_out/r03/_m.json'))['cur']
print('foods %2d  %-12s %-16s mean %.4f ms  min %.4f  max %.4f' % ($F, '$SIG' or 'FULL', '$TANK' or 'std constants', d['mean_ms'], d['min_ms'], d['max_ms']))"
    done
  done
done
