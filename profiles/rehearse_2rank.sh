set -e
mkdir -p gpurun_out/r02
SALP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 1 --total-envs 32768 --chunk 100 > gpurun_out/r02/bench_2rank_rehearsal.json 2> gpurun_out/r02/bench_2rank_rehearsal.err
python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_2rank_rehearsal.json').read().strip().splitlines()[-1]); print('rehearsal', d['value'], d['other_exchange_modes'])"
