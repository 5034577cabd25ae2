cd $GRAFT_REPO_ROOT
bash scripts/gpu_prof_train.sh r06_train scripts/train_only.py > gpurun_out/r06_train.log 2>&1
bash scripts/gpu_prof_train.sh r06_train14 scripts/train14_only.py > gpurun_out/r06_train14.log 2>&1
bash scripts/gpu_prof_train.sh r06_train14_3010 scripts/train14_3010_only.py > gpurun_out/r06_train14_3010.log 2>&1
for t in r06_train r06_train14 r06_train14_3010; do
  f=$(find gpurun_out/$t/trace -name "*kernel_trace.csv" | head -1)
  python scripts/step_timeline.py $f sgd_momentum_kernel > gpurun_out/${t}_timeline.txt 2>&1
  find gpurun_out/$t -name "*kernel_trace.csv" -size +20M -delete
  head -3 gpurun_out/$t/summary_step_breakdown.txt
done
