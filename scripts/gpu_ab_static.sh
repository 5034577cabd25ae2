cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
CHECK=1 KNOB=21 VALUES=0,2,3 NETS=8 FRAMES=49,98 ROUNDS=3 timeout 400 python scripts/ab_knob_train.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4s/ab_conv0c.log
