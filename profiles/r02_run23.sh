set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_v14.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v14.log; exit 1; }
tail -2 gpurun_out/r02/gpu_tests_v14.log
SALP_HIP_LIBRARY=$PWD/profiles/ab/v14_w4.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_full_horizon.py -m gpu -x -q > gpurun_out/r02/gpu_tests_v14w4.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v14w4.log; exit 1; }
tail -2 gpurun_out/r02/gpu_tests_v14w4.log
timeout -k 10 300 python profiles/ab_bench.py v12=profiles/ab/v12.so v14=profiles/ab/v14.so v14w4=profiles/ab/v14_w4.so --preset sac_gail --rounds 8 > gpurun_out/r02/ab_sacgail_14.json 2>gpurun_out/r02/ab_sacgail_14.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_14.json')); print('F12', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
timeout -k 10 300 python profiles/ab_bench.py v12=profiles/ab/v12.so v14=profiles/ab/v14.so v14w4=profiles/ab/v14_w4.so --rounds 6 > gpurun_out/r02/ab_f1_4.json 2>gpurun_out/r02/ab_f1_4.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_f1_4.json')); print('F1', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
