set -e
mkdir -p gpurun_out/r02
timeout -k 10 300 python profiles/ab_bench.py v15=profiles/ab/v15.so ntsmall=profiles/ab/v15_ntsmall.so nodrain=profiles/ab/v15_nodrain.so v15+gen=profiles/ab/v15.so --rounds 8 > gpurun_out/r02/ab_f1_5.json 2>gpurun_out/r02/ab_f1_5.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_f1_5.json')); print('F1', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
