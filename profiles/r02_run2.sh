set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_vreg.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_vreg.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests_vreg.log
timeout -k 10 300 python profiles/ab_bench.py r01=profiles/ab/r01.so ldsfood=profiles/ab/ldsfood.so vreg=profiles/ab/vreg.so --preset sac_gail > gpurun_out/r02/ab_sacgail_2.json 2>gpurun_out/r02/ab_sacgail_2.err
cat gpurun_out/r02/ab_sacgail_2.json
