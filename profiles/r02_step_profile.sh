#!/bin/bash
# H = 1 (step-per-launch) profile of salp_vec_step at 262144 and 4096 envs: kernel trace + FETCH / WRITE PMC passes.
set -e
export TMPDIR=/tmp
O=gpurun_out/r02/step
mkdir -p $O
python3 profiles/step_mode.py 4096 262144 > $O/step_mode.jsonl
cat $O/step_mode.jsonl
for n in 4096 262144; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 profiles/step_mode.py $n > /dev/null 2> $O/kt_$n.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$n -- python3 profiles/step_mode.py $n > /dev/null 2> $O/pf_$n.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$n -- python3 profiles/step_mode.py $n > /dev/null 2> $O/pw_$n.err
  python3 - <<PY
import csv, glob, collections, json
n = $n
ks = glob.glob("$O/kt_%d/*/*_kernel_stats.csv" % n)
row = [r for r in csv.DictReader(open(ks[0])) if 'salp_rollout_kernel' in r['Name']][0]
out = {"envs": n, "kernel": row['Name'][:80], "calls": int(row['Calls']), "avg_ns": float(row['AverageNs']), "min_ns": float(row['MinNs'])}
for c in ("fetch", "write"):
    v = []
    for f in glob.glob("$O/pmc_%s_%d/*/*_counter_collection.csv" % (c, n)):
        for r in csv.DictReader(open(f)):
            if 'salp_rollout_kernel' in r['Kernel_Name']: v.append(float(r['Counter_Value']))
    out[c + "_kb_per_launch"] = sum(v) / max(len(v), 1)
bpe = 218.0
out["algorithmic_bytes"] = bpe * n
out["achieved_GBps"] = bpe * n / out["avg_ns"]
out["frac_of_8TBps"] = out["achieved_GBps"] / 8000.0
out["pmc_bytes_raw"] = (out["fetch_kb_per_launch"] + out["write_kb_per_launch"]) * 1024
print(json.dumps(out))
open("$O/step_profile_%d.json" % n, "w").write(json.dumps(out, indent=1))
import shutil; shutil.copy(ks[0], "$O/step_%d_kernel_stats.csv" % n)
PY
done
