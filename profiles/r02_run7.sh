set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_v3.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v3.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests_v3.log
timeout -k 10 300 python profiles/ab_bench.py r01=profiles/ab/r01.so nohoist=profiles/ab/nohoist.so v3=profiles/ab/v3.so --preset sac_gail > gpurun_out/r02/ab_sacgail_4.json 2>gpurun_out/r02/ab_sacgail_4.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_4.json')); print({k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
bash profiles/pmc_wave.sh v3_4w profiles/ab/v3.so sac_gail 262144 > gpurun_out/r02/pmcw_v3_4w.txt
grep -E "INSTS_VALU |INSTS_SALU|ACTIVE_INST_VALU|WAVE_CYCLES|GRBM|INSTS_LDS|BRANCH|WAIT_ANY|WAIT_INST_ANY" gpurun_out/r02/pmcw_v3_4w.txt
