set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_v6.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_v6.log; exit 1; }
tail -3 gpurun_out/r02/gpu_tests_v6.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_n1_quick.json 2> gpurun_out/r02/bench_n1_quick.err
cat gpurun_out/r02/bench_n1_quick.json
SALP_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_forced_sharded.json 2> gpurun_out/r02/bench_forced_sharded.err
cat gpurun_out/r02/bench_forced_sharded.json
SALP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --total-envs 32768 --chunk 100 > gpurun_out/r02/bench_2rank_rehearsal.json 2> gpurun_out/r02/bench_2rank_rehearsal.err
cat gpurun_out/r02/bench_2rank_rehearsal.json
