#!/bin/bash
# effective shader clock per variant: GRBM_GUI_ACTIVE (sum over 8 XCDs) / 8 / dispatch duration
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/clock_probe; rm -rf $OUT; mkdir -p $OUT
for v in "$@"; do
  AB_WARM=6 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/$v -- python3 profiles/ab_bench.py $v=profiles/ab/$v.so --rounds 2 --launches 4 > $OUT/$v.json 2> $OUT/$v.err
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for v in sys.argv[2:]:
    f = glob.glob(f"{out}/{v}/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if 'rollout' in r['Kernel_Name'] and r['Counter_Name']=='GRBM_GUI_ACTIVE']
    rows = rows[-8:]
    clk = [float(r['Counter_Value'])/8/((int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-9)/1e9 for r in rows]
    dur = [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-6 for r in rows]
    print(f"{v:12s} clock GHz {sum(clk)/len(clk):.3f}  kernel ms {sum(dur)/len(dur):.3f}  cycles/launch {sum(clk)/len(clk)*sum(dur)/len(dur)*1e6:.3e}")
PY
