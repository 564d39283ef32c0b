#!/bin/bash
# rocprofv3 kernel statistics of bench.py itself (the command whose JSON line is the headline):
#   scripts/prof_bench.sh <tag> [bench.py args]  -> gpurun_out/prof_<tag>/stats.csv, bench.json
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-calibration "$@" > $out/bench.json 2> $out/log.txt
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $out/stats.csv
head -12 $out/stats.csv
