#!/bin/bash
# usage: scripts/prof_any.sh <tag> <script.py> [args...]  -> gpurun_out/prof_<tag>/stats.csv (rocprofv3 kernel statistics)
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $script "$@" > $out/log.txt 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $out/stats.csv
tail -1 $out/log.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot/1e6)
for r in rows[:26]:
    print(f'{r["Name"][:74]:74s} {int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):6.2f}%')
PY
