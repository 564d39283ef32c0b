#!/bin/bash
# rocprofv3 kernel statistics of scripts/train_bn.py -> gpurun_out/prof_bn/stats.csv
set -e
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_bn
rm -rf $out; mkdir -p $out
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/scripts/train_bn.py > $out/log.txt 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $out/stats.csv
tail -1 $out/log.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot/1e6)
for r in rows[:30]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):6.2f}%')
PY
