set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_v10.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v10.log; exit 1; }
tail -2 gpurun_out/r02/gpu_tests_v10.log
timeout -k 10 300 python profiles/ab_bench.py r01=profiles/ab/r01.so v5=profiles/ab/v5.so v10=profiles/ab/v10.so > gpurun_out/r02/ab_f1_3.json 2>gpurun_out/r02/ab_f1_3.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_f1_3.json')); print('F1', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
timeout -k 10 300 python profiles/ab_bench.py r01=profiles/ab/r01.so v8=profiles/ab/v8.so v10=profiles/ab/v10.so --preset sac_gail > gpurun_out/r02/ab_sacgail_10.json 2>gpurun_out/r02/ab_sacgail_10.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_10.json')); print('F12', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
timeout -k 10 200 python examples/train_sac_gail.py --steps 300 > gpurun_out/r02/example_sac_gail.log 2>&1 || tail -5 gpurun_out/r02/example_sac_gail.log
tail -c 600 gpurun_out/r02/example_sac_gail.log
python -c "import __graft_entry__ as g; g.smoke()"
