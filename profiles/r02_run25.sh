set -e
mkdir -p gpurun_out/r02
for v in v14 robot_nolicm; do
  echo "== $v"
  SALP_HIP_LIBRARY=$PWD/profiles/ab/$v.so timeout -k 10 300 python3 profiles/robot_perf.py 2>/dev/null | grep '"schedule": 1'
done | tee gpurun_out/r02/robot_licm_ab.txt
