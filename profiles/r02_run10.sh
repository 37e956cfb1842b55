set -e
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_v5.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v5.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests_v5.log
timeout -k 10 300 python profiles/ab_bench.py r01=profiles/ab/r01.so padtile=profiles/ab/padtile.so swz=profiles/ab/v5.so > gpurun_out/r02/ab_f1_2.json 2>gpurun_out/r02/ab_f1_2.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_f1_2.json')); print({k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
timeout -k 10 300 python profiles/ab_bench.py padtile=profiles/ab/padtile.so swz=profiles/ab/v5.so --preset sac_gail > gpurun_out/r02/ab_sacgail_6.json 2>gpurun_out/r02/ab_sacgail_6.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_sacgail_6.json')); print({k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
for v in padtile v5; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/r02/pmc_lds_$v -- python3 profiles/ab_bench.py x=profiles/ab/$v.so --rounds 1 --launches 3 > /dev/null 2> gpurun_out/r02/pmc_lds_$v.err
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r02/pmc_lds_$v/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'salp_rollout' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("$v", {k: round(sum(x)/len(x)) for k,x in agg.items()}, 'launches', {k:len(x) for k,x in agg.items()})
PY
done | tee gpurun_out/r02/pmc_lds_conflicts.txt
