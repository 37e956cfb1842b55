set -e
mkdir -p gpurun_out/r02
timeout -k 10 400 python profiles/ab_bench.py base=underwater-swimmer_rl_amd/csrc/libsalp_hip.so sc0sc1nt=profiles/ab/bits_a.so sc1nt=profiles/ab/bits_b.so sc0nt=profiles/ab/bits_c.so sc1=profiles/ab/bits_d.so sc0sc1=profiles/ab/bits_e.so --rounds 6 > gpurun_out/r02/ab_store_bits.json 2>gpurun_out/r02/ab_store_bits.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab_store_bits.json')); print('F1', {k:(round(v['median_ms'],4),round(v['min_ms'],4)) for k,v in d.items()})"
