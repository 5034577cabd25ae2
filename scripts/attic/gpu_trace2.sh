#!/bin/bash
# kernel traces of the TCResNet8 / TCResNet14-1.5 training steps (scripts/train_only.py, train14_only.py) -> median-step timelines
# usage: gpurun -- 'bash scripts/gpu_trace2.sh tag [TUNE string, e.g. 7=1]'
TAG=$1
export TUNE=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for w in train_only train14_only; do
  STEPS=12 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$w -o t -- python $R/scripts/$w.py > $OUT/$w.log 2>&1; echo "$w rc=$?"
  f=$(find $OUT/$w -name '*kernel_trace.csv' | head -1)
  python $R/scripts/step_timeline.py $f sgd_momentum_kernel > $OUT/${w}_timeline.txt
  wc -l $OUT/${w}_timeline.txt
done
