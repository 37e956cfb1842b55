set -e
mkdir -p gpurun_out/r02
for n in 65536 131072 196608 262144; do
  echo "envs $n"
  timeout -k 10 300 python profiles/ab_bench.py vreg=profiles/ab/vreg.so nodrain=profiles/ab/vreg_nodrain.so vm6=profiles/ab/vreg_vm6.so ldsfood=profiles/ab/ldsfood.so --preset sac_gail --envs $n --rounds 4 2>gpurun_out/r02/ab_occ.err | python -c "
import json,sys; d=json.load(sys.stdin); print({k: round(v['median_ms'],4) for k,v in d.items()})"
done | tee gpurun_out/r02/ab_occupancy_curve.txt
