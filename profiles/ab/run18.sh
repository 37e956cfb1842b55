set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/v18_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r02/v18_gpu_tests.log; exit 1; }
tail -1 gpurun_out/r02/v18_gpu_tests.log
timeout -k 10 400 python profiles/ab_bench.py v17=profiles/ab/v17.so v18=profiles/ab/v18.so --preset sac_gail --rounds 8 > gpurun_out/r02/ab_sacgail_18.json 2>gpurun_out/r02/ab_sacgail_18.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_18.json')); print('F12', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
timeout -k 10 400 python profiles/ab_bench.py v17=profiles/ab/v17.so v18=profiles/ab/v18.so --preset defaults --rounds 6 > gpurun_out/r02/ab_defaults_18.json 2>gpurun_out/r02/ab_defaults_18.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_defaults_18.json')); print('F5', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
