#!/bin/bash
# Ablation sweep of wgrad_f32_kernel (variant libraries dbgvar/wg_dbg<bits>.so) under rocprofv3.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1 2 3 4 7; do
  for c in "128 64 128 2" "64 128 256 1"; do
    set -- $c
    if [ $v = 0 ]; then unset ERFNET_HIP_SO; else export ERFNET_HIP_SO=$R/dbgvar/wg_dbg$v.so; fi
    rm -rf /tmp/pp
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $R/scripts/micro_nb1d.py --c $1 --h $2 --w $3 --d $4 --reps 10 --bwd > /tmp/pp.log 2>&1 || { echo "variant $v failed"; tail -5 /tmp/pp.log; exit 1; }
    mkdir -p $R/gpurun_out/wg; cp /tmp/pp/p_kernel_stats.csv $R/gpurun_out/wg/dbg${v}_c$1.csv; echo "dbg=$v C=$1 ok"
  done
done
