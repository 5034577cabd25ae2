#!/bin/bash
# Training-step profile: rocprofv3 kernel trace + HBM-traffic PMC passes of scripts/train_only.py (TCResNet8-1.0, batch 4096,
# features precomputed).  usage: gpurun -- 'bash scripts/gpu_prof_train.sh tag'; then scripts/train_breakdown.py condenses it.
TAG=${1:-r02train}
SCRIPT=${2:-scripts/train_only.py}      # profiling target
MARKER=${3:-sgd_momentum_kernel}        # kernel that ends a step
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export STEPS=${STEPS:-8}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/$SCRIPT > $OUT/trace.log 2>&1; echo "trace rc=$?"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- python $R/$SCRIPT > $OUT/pmc$i.log 2>&1; echo "pmc pass $i rc=$?"
done
python $R/scripts/train_breakdown.py $OUT $OUT/summary $MARKER
