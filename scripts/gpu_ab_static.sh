cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "static_phase or every_kernel_path or eval_forward or test_train or full_batch or pipeline or staged" > gpurun_out/r4s/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4s/pytest.log
timeout 300 python scripts/ab_knob_fwd.py > gpurun_out/r4s/ab_fwd.log 2>&1; cat gpurun_out/r4s/ab_fwd.log
KNOB=19 VALUES=0,1,2,3 NETS=8,14 FRAMES=49,98 timeout 400 python scripts/ab_knob_train.py > gpurun_out/r4s/ab_train.log 2>&1; cat gpurun_out/r4s/ab_train.log
