cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4s/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4s/pytest_all.log
