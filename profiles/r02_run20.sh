set -e
mkdir -p gpurun_out/r02
SALP_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_forced_sharded.json 2> gpurun_out/r02/bench_forced_sharded.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_forced_sharded.json').read().strip().splitlines()[-1]); print('forced sharded', d['value'], d['ms_per_step'], d['kernel_side_value'])"
SALP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --total-envs 32768 --chunk 100 > gpurun_out/r02/bench_2rank_rehearsal.json 2> gpurun_out/r02/bench_2rank_rehearsal.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_2rank_rehearsal.json').read().strip().splitlines()[-1]); print('rehearsal', d['value'], d['exchange'])"
SALP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 2 --warmup 1 --total-envs 131072 --chunk 250 > gpurun_out/r02/bench_2rank_rehearsal_b.json 2> gpurun_out/r02/bench_2rank_rehearsal_b.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_2rank_rehearsal_b.json').read().strip().splitlines()[-1]); print('rehearsal b', d['value'], d['exchange']['recv_bytes_per_rank_per_launch'])"
