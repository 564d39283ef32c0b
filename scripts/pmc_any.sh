#!/bin/bash
# PMC passes (one rocprofv3 run per counter set) over a micro-benchmark script.
# usage: scripts/pmc_any.sh <tag> <script.py> <args...>   -> gpurun_out/pmc_<tag>/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
script=$1; shift
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVES" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o pass$i -- python3 $script "$@" > gpurun_out/pmc_${tag}_pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmc_${tag}_pass$i.log; [ $i -lt 5 ] && exit 1; }
done
echo done $tag
