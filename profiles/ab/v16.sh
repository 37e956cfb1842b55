set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_v16.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v16.log; exit 1; }
tail -1 gpurun_out/r02/gpu_tests_v16.log
timeout -k 10 400 python profiles/ab_bench.py r01=profiles/ab/r01.so nt=profiles/ab/v16_nt.so sc1nt=profiles/ab/v16.so --rounds 8 > gpurun_out/r02/ab_f1_6.json 2>gpurun_out/r02/ab_f1_6.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_f1_6.json')); print('F1', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
timeout -k 10 400 python profiles/ab_bench.py nt=profiles/ab/v16_nt.so sc1nt=profiles/ab/v16.so --preset sac_gail --rounds 8 > gpurun_out/r02/ab_sacgail_16.json 2>gpurun_out/r02/ab_sacgail_16.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_16.json')); print('F12', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
