#!/bin/bash
# kernel trace only (sqlite db) of the forward bench; usage: bash scripts/gpu_trace.sh tag
TAG=${1:-t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline ${2:-} > $OUT/prof.log 2>&1; echo "rocprof rc=$?"
