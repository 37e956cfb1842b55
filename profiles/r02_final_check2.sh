set -e
mkdir -p gpurun_out/r02
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r02/final_bench.json 2> gpurun_out/r02/final_bench.err
echo "bench wall seconds: $SECONDS"
python -c "
import json; d=json.loads(open('gpurun_out/r02/final_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['avg_kernel_ms'], d['config']['workload'][:40], d['cpu_baseline']['value'], d['cpu_baseline']['python_reference']['value'], d['sac_first_capture'])"
