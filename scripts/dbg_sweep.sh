#!/bin/bash
# ablation sweep over a diagnostic build (dbgvar/lib_dbg.so): ERFNET_WGRAD_DBG bits
# 1 = no slab stores, 2 = no global loads in the loop, 4 = no LDS writes, 8 = no MFMAs
export ERFNET_HIP_SO=$GRAFT_REPO_ROOT/dbgvar/lib_dbg.so
export ERFNET_NO_SIDE_STREAM=1
for c in "--c 128 --h 64 --w 128 --d 2" "--c 64 --h 128 --w 256 --d 1"; do
for f in 0 1 2 4 6 7 8 14 15; do
  echo -n "dbg=$f  "
  ERFNET_WGRAD_DBG=$f timeout -k 10 100 python scripts/micro_nb1d.py $c --reps 20 --bwd || exit 1
done; done
