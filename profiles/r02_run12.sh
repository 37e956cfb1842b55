set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_sac_gail.py tests/test_gail_parity.py -m gpu -x -q -s > gpurun_out/r02/gpu_tests_sac.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_sac.log; exit 1; }
grep -E "configs\[4\]|passed|failed" gpurun_out/r02/gpu_tests_sac.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_n1_probe.json 2> gpurun_out/r02/bench_n1_probe.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_n1_probe.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d.get('sac_first_capture'))"
