#!/bin/bash
# One gpurun call: smoke + GPU parity tests + bench + rocprofv3 kernel trace.  Logs land in gpurun_out/.
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_check.sh [tag]'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
echo "== pytest -m gpu"; timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
echo "== bench"; timeout 400 python bench.py --steps 30 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprofv3 kernel trace"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline > $OUT/prof.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*" | head -3
for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -40 $f; done
# keep the merge small: drop the big per-dispatch trace, keep stats
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
rocm-smi --showproductname 2>/dev/null | head -8
lscpu | grep -E "Model name|^CPU\(s\)" | head -3
