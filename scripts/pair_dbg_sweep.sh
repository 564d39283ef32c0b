#!/bin/bash
# Ablation sweep of pair_f32_kernel: each variant library (dbgvar/pair_dbg<bits>.so) is run under
# rocprofv3 on the 128-channel and 64-channel training micro-benchmark; prints the average kernel time.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1 2 3 4 8 12 15 31; do
  for c in "128 64 128 2" "64 128 256 1"; do
    set -- $c
    if [ $v = 0 ]; then unset ERFNET_HIP_SO; else export ERFNET_HIP_SO=$R/dbgvar/pair_dbg$v.so; fi
    rm -rf /tmp/pp
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $R/scripts/micro_nb1d.py --c $1 --h $2 --w $3 --d $4 --reps 10 --bwd > /tmp/pp.log 2>&1 || { echo "variant $v failed"; tail -5 /tmp/pp.log; exit 1; }
    echo "dbg=$v C=$1: $(grep pair_f32 /tmp/pp/p_kernel_stats.csv | cut -d, -f1-4 | tr '\n' ' ')"
  done
done
