#!/bin/bash
# two independent training processes on one GPU at the same time (no process group): do they both finish?
cd $GRAFT_REPO_ROOT
timeout -k 10 100 python3 scripts/prof_step.py train f32 > gpurun_out/s9_two_a.log 2>&1 &
pa=$!
timeout -k 10 100 python3 scripts/prof_step.py train f32 > gpurun_out/s9_two_b.log 2>&1 &
pb=$!
wait $pa; ra=$?
wait $pb; rb=$?
echo "exit codes: $ra $rb"
[ $ra -eq 0 ] && [ $rb -eq 0 ]
