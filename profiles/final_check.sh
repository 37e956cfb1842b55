set -e
mkdir -p gpurun_out/r02
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r02/final_build.log 2>&1 || { tail -20 gpurun_out/r02/final_build.log; exit 1; }
tail -1 gpurun_out/r02/final_build.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/final_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r02/final_gpu_tests.log; exit 1; }
tail -1 gpurun_out/r02/final_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | cut -c1-120
SECONDS=0
python bench.py > gpurun_out/r02/final_bench.json 2> gpurun_out/r02/final_bench.err
echo "bench wall seconds: $SECONDS"
python -c "
import json; d=json.loads(open('gpurun_out/r02/final_bench.json').read().strip().splitlines()[-1]); print(d['value'], round(d['roofline']['frac'],4), d['sac_first_capture']['vector_steps'])"
