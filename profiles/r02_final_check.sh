set -e
mkdir -p gpurun_out/r02
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r02/final_build.log 2>&1 || { tail -20 gpurun_out/r02/final_build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/final_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r02/final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r02/final_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
/usr/bin/time -v python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r02/final_bench.json 2> gpurun_out/r02/final_bench.err
grep -E "Elapsed|Maximum resident" gpurun_out/r02/final_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/final_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['avg_kernel_ms'], d['config']['workload'][:40], d['cpu_baseline']['value'], d['cpu_baseline']['python_reference']['value'], d['sac_first_capture'])"
