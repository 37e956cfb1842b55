set -e
mkdir -p gpurun_out/r02
for n in 65536 131072 196608 262144; do
  echo "envs $n"
  timeout -k 10 300 python profiles/ab_bench.py vreg_nodrain=profiles/ab/vreg_nodrain.so nohoist=profiles/ab/nohoist.so hoist=profiles/ab/hoist.so --preset sac_gail --envs $n --rounds 4 2>gpurun_out/r02/ab_occ2.err | python -c "
import json,sys; d=json.load(sys.stdin); print({k: round(v['median_ms'],4) for k,v in d.items()})"
done | tee gpurun_out/r02/ab_occupancy_curve2.txt
