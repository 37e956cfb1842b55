set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_v7.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v7.log; exit 1; }
tail -2 gpurun_out/r02/gpu_tests_v7.log
timeout -k 10 300 python profiles/ab_bench.py v4=profiles/ab/v4.so v5=profiles/ab/v5.so v7=profiles/ab/v7.so --preset sac_gail > gpurun_out/r02/ab_sacgail_7.json 2>gpurun_out/r02/ab_sacgail_7.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_7.json')); print({k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
