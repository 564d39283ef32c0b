cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_wg -o pass$i -- python scripts/micro_nb1d.py --c 128 --h 64 --w 128 --d 2 --reps 5 --bwd > gpurun_out/pmc_wg_pass$i.log 2>&1 || { echo "pass $i failed"; exit 1; }
done
echo done
