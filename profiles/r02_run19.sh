set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_final.log 2>&1 || { tail -40 gpurun_out/r02/gpu_tests_final.log; exit 1; }
tail -2 gpurun_out/r02/gpu_tests_final.log
SALP_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_forced_sharded.json 2> gpurun_out/r02/bench_forced_sharded.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_forced_sharded.json').read().strip().splitlines()[-1]); print('forced sharded', d['value'], d['exchange'], d['kernel_side_value'])"
SALP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --total-envs 32768 --chunk 100 > gpurun_out/r02/bench_2rank_rehearsal.json 2> gpurun_out/r02/bench_2rank_rehearsal.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_2rank_rehearsal.json').read().strip().splitlines()[-1]); print('rehearsal', d['value'], d['config']['workload'][:60], d['exchange']['mode'])"
bash profiles/profile.sh r02_g_final > gpurun_out/r02/profile_g_final.log 2>&1
tail -1 gpurun_out/r02/profile_g_final.log | cut -c1-400
bash profiles/profile.sh r02_g_sacgail --preset sac_gail > gpurun_out/r02/profile_g_sacgail.log 2>&1
tail -1 gpurun_out/r02/profile_g_sacgail.log | cut -c1-400
