set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_v8.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v8.log; exit 1; }
tail -2 gpurun_out/r02/gpu_tests_v8.log
SALP_HIP_LIBRARY=$PWD/profiles/ab/v8_w4.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/r02/gpu_tests_v8w4.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v8w4.log; exit 1; }
tail -2 gpurun_out/r02/gpu_tests_v8w4.log
timeout -k 10 300 python profiles/ab_bench.py v7=profiles/ab/v7.so v8=profiles/ab/v8.so v8w4=profiles/ab/v8_w4.so --preset sac_gail > gpurun_out/r02/ab_sacgail_8.json 2>gpurun_out/r02/ab_sacgail_8.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_8.json')); print({k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
timeout -k 10 300 python profiles/ab_bench.py v7=profiles/ab/v7.so v8=profiles/ab/v8.so v8w4=profiles/ab/v8_w4.so --preset sac_gail --envs 1048576 --rounds 3 > gpurun_out/r02/ab_sacgail_8_1m.json 2>gpurun_out/r02/ab_sacgail_8_1m.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_8_1m.json')); print('1M envs', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
