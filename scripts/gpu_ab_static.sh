cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
CHECK=1 KNOB=22 VALUES=1,0 NETS=8 FRAMES=49,98 ROUNDS=3 timeout 600 python scripts/ab_knob_train.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4s/ab_vecstage.log
CHECK=1 TUNE=9=3 KNOB=22 VALUES=1,0 NETS=14 FRAMES=49 ROUNDS=2 timeout 600 python scripts/ab_knob_train.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4s/ab_vecstage.log
