#!/bin/bash
# ablation sweep over a diagnostic build (dbgvar/lib_dbg.so): ERFNET_IGEMM_DBG bits
# 1 = no output stores, 2 = no global loads in the loop, 4 = no LDS writes, 8 = no MFMAs, 16 = no epilogue
export ERFNET_HIP_SO=$GRAFT_REPO_ROOT/dbgvar/lib_dbg.so
for c in "--c 128 --h 64 --w 128 --d 2" "--c 64 --h 128 --w 256 --d 1"; do
for f in 0 1 16 6 8 14 22 30; do
  echo -n "dbg=$f  "
  ERFNET_IGEMM_DBG=$f timeout -k 10 100 python scripts/micro_nb1d.py $c --reps 20 || exit 1
done; done
