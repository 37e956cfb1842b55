set -e
mkdir -p gpurun_out/r02
export SALP_HIP_LIBRARY=$PWD/profiles/ab/nolicm.so
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_nolicm.log 2>&1 || { tail -30 gpurun_out/r02/gpu_tests_nolicm.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests_nolicm.log
unset SALP_HIP_LIBRARY
timeout -k 10 300 python profiles/ab_bench.py r01=profiles/ab/r01.so cur=profiles/ab/cur.so nolicm=profiles/ab/nolicm.so nothrust=profiles/ab/nolicm_nothrust.so --preset sac_gail > gpurun_out/r02/ab_sacgail_1.json 2>gpurun_out/r02/ab_sacgail_1.err
cat gpurun_out/r02/ab_sacgail_1.json
AB_WARM=0 AB_DUMP=1 timeout -k 10 300 python profiles/ab_bench.py nolicm=profiles/ab/nolicm.so --preset sac_gail --rounds 1 --launches 24 > gpurun_out/r02/ab_sacgail_warm0.json 2>gpurun_out/r02/ab_sacgail_warm0.err
cat gpurun_out/r02/ab_sacgail_warm0.json
timeout -k 10 300 python profiles/ab_bench.py r01=profiles/ab/r01.so cur=profiles/ab/cur.so nolicm=profiles/ab/nolicm.so > gpurun_out/r02/ab_f1_1.json 2>gpurun_out/r02/ab_f1_1.err
cat gpurun_out/r02/ab_f1_1.json
