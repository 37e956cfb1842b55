set -e
mkdir -p gpurun_out/r02
timeout -k 10 300 python profiles/ab_bench.py nohoist=profiles/ab/nohoist.so v3=profiles/ab/v3.so noxor3=profiles/ab/v3_noxor3.so ieeediv=profiles/ab/v3_ieeediv.so neither=profiles/ab/v3_neither.so --preset sac_gail > gpurun_out/r02/ab_sacgail_5.json 2>gpurun_out/r02/ab_sacgail_5.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_5.json')); print({k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
