#!/bin/bash
# A/B of environment switches on one box: scripts/ab_env.sh "VAR=1" "" ... runs bench.py once per
# setting (an empty string = defaults), twice round-robin, and prints images/s per run.
# AB_ARGS: further bench.py arguments (e.g. "--dtype bf16 --steps 50").
args="--no-extras --no-cpu-baseline"
for round in 1 2; do
  for setting in "$@"; do
    v=$(env $setting timeout -k 10 200 python bench.py $args $AB_ARGS 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'])") || exit 1
    echo "[$setting]: $v"
  done
done
