#!/bin/bash
# runs scripts/x3_micro.py under rocprofv3 for the library given in ERFNET_HIP_SO and prints the line_x3 kernel rows
for so in "$@"; do
  export ERFNET_HIP_SO=$GRAFT_REPO_ROOT/$so
  echo "== $so"
  bash $GRAFT_REPO_ROOT/scripts/prof_any.sh xe scripts/x3_micro.py | grep "line_x3" || exit 1
done
