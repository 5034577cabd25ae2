cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_train or eval_forward or full_batch or small_batches or wide_net or paths_agree" > gpurun_out/r4s/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4s/pytest.log
KNOB=20 VALUES=0 NETS=8,14 FRAMES=49,98 ROUNDS=3 timeout 400 python scripts/ab_knob_train.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4s/ab_head.log
