#!/bin/bash
# usage: scripts/bench_quick.sh [steps]  -> value ms_per_step roofline-us   (fp32 headline, no extras)
steps=${1:-20}
timeout -k 10 280 python3 $GRAFT_REPO_ROOT/bench.py --steps $steps --warmup 5 --no-extras --no-cpu-baseline > /tmp/bq.json 2> /tmp/bq.err || { tail -3 /tmp/bq.err; exit 1; }
python3 -c "
import json;d=json.load(open('/tmp/bq.json'));print(d['value'],d['ms_per_step'],d['roofline']['us_per_launch'])"
