#!/bin/bash
# One gpurun call: smoke + every GPU test + bench (+ optional profile pack).  usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh tag [prof]'
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest_gpu.log
echo "== bench"; timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
if [ "$2" = "prof" ]; then bash scripts/gpu_prof.sh $TAG/prof; fi
rocm-smi --showproductname 2>/dev/null | head -6
lscpu | grep -E "Model name|^CPU\(s\)" | head -3
