#!/bin/bash
# rocprofv3 kernel-trace stats (CSV) + PMC passes for the bench command.  usage: gpurun -- 'bash scripts/gpu_prof.sh tag'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 -L > $OUT/counters_list.txt 2>&1
echo "== kernel trace"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1; echo rc=$?
echo "== kernel trace, the one-stream forward sequence (the roofline's kernel_ms -- the kernels alone on the chip -- is recomputable from this one)"
FWD="python $R/scripts/fwd_seq_only.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_fwd -o t -- $FWD > $OUT/trace_fwd.log 2>&1; echo rc=$?
find $OUT/trace $OUT/trace_fwd -name "*kernel_trace.csv" -size +30M -delete
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "== pmc pass $i: $PMC"
  STEPS=20 timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- $FWD > $OUT/pmc$i.log 2>&1; echo rc=$?
done
ls -R $OUT | head -40
