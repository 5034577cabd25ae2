#!/bin/bash
# rocprofv3 kernel trace (sqlite) of an arbitrary python command.  usage: gpurun -- 'bash scripts/gpu_trace.sh tag <python args...>'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python "$@" > $OUT/run.log 2>&1; echo "rc=$?"
grep -v amdgpu.ids $OUT/run.log | tail -20
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
