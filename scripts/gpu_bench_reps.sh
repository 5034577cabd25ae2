#!/bin/bash
# bench.py with the driver's flags, REPS times per pipeline width: how the K = 20 timed region's fill / drain shows in the headline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/reps
for rep in $(seq 1 ${REPS:-5}); do
  for w in 3 2; do
    TCR_BENCH_WAYS=$w timeout 200 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --legs none 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ways $w rep $rep', o['ms_per_step'], o['sequential']['ms_per_step'])"
  done
done
