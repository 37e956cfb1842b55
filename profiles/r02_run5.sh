set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_hoist.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_hoist.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests_hoist.log
timeout -k 10 300 python profiles/ab_bench.py r01=profiles/ab/r01.so vreg_nodrain=profiles/ab/vreg_nodrain.so nohoist=profiles/ab/nohoist.so hoist=profiles/ab/hoist.so --preset sac_gail > gpurun_out/r02/ab_sacgail_3.json 2>gpurun_out/r02/ab_sacgail_3.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_3.json')); print({k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
