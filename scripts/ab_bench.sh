#!/bin/bash
# A/B of library builds on one box: scripts/ab_bench.sh [bench args --] variants/lib_a.so variants/lib_b.so ...
# runs bench.py once per library, twice round-robin, and prints images/s per run
args="--no-extras --no-cpu-baseline"
for round in 1 2; do
  for so in "$@"; do
    v=$(ERFNET_HIP_SO=$GRAFT_REPO_ROOT/$so timeout -k 10 200 python bench.py $args $AB_ARGS 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'])") || exit 1
    echo "$so: $v"
  done
done
