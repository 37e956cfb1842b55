#!/bin/bash
# usage (on the GPU box, via gpurun): bash profiles/ab.sh <tag> <preset> <rounds> name=path.so ...
# Interleaved A/B of library builds (profiles/ab_bench.py); JSON to gpurun_out/r03/ab_<tag>.json, one line per variant on stdout.
TAG=$1; PRESET=$2; ROUNDS=$3; shift 3
mkdir -p gpurun_out/r03
python profiles/ab_bench.py "$@" --preset $PRESET --rounds $ROUNDS --launches 5 2> gpurun_out/r03/ab_$TAG.err > gpurun_out/r03/ab_$TAG.json || { tail -5 gpurun_out/r03/ab_$TAG.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/r03/ab_$TAG.json"))
for k, v in d.items():
    print(f"{k:12s} mean {v['mean_ms']:.4f} median {v['median_ms']:.4f} min {v['min_ms']:.4f} max {v['max_ms']:.4f} paired {v['paired_ratio_to_first']:.4f}")
PY
