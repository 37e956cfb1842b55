#!/bin/bash
# On the GPU box: the other template axes of the rollout kernel — free breathing (act_dim 2) and in-kernel actions — per food count.
LIB=${1:-underwater-swimmer_rl_amd/csrc/libsalp_hip.so}
mkdir -p gpurun_out/r03
run() { # label, variant-name, extra args...
  L=$1; V=$2; shift 2
  python profiles/ab_bench.py $V=$LIB --preset sac_gail "$@" --rounds 2 --launches 5 > gpurun_out/r03/_m.json 2> gpurun_out/r03/_m.err || { tail -3 gpurun_out/r03/_m.err; exit 1; }
  python -c "
import json; d=list(json.load(open('gpurun_out/r03/_m.json')).values())[0]
print('%-60s mean %.4f ms  min %.4f  max %.4f' % ('$L', d['mean_ms'], d['min_ms'], d['max_ms']))"
}
for F in 1 5 12 16; do
  run "foods $F forced FULL" cur --set num_food_items=$F
  run "foods $F free-breathing FULL" cur --set num_food_items=$F --set forced_breathing=false
  run "foods $F free-breathing final-obs" cur --set num_food_items=$F --set forced_breathing=false --final-obs
  run "foods $F forced, actions in kernel, written out" cur+gen --set num_food_items=$F
  run "foods $F forced, actions in kernel, not written" cur+gennoout --set num_food_items=$F
done
