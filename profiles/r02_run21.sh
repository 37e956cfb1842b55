set -e
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_full_horizon.py tests/test_gpu_vector_env.py -m gpu -x -q > gpurun_out/r02/gpu_tests_v12.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v12.log; exit 1; }
tail -2 gpurun_out/r02/gpu_tests_v12.log
timeout -k 10 300 python profiles/ab_bench.py v11=profiles/ab/v11.so v12=profiles/ab/v12.so --preset sac_gail --rounds 8 > gpurun_out/r02/ab_sacgail_12.json 2>gpurun_out/r02/ab_sacgail_12.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_12.json')); print({k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
