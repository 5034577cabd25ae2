cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
CHECK=1 KNOB=20 VALUES=0,8,12,16 NETS=8 FRAMES=49,98 ROUNDS=3 timeout 400 python scripts/ab_knob_train.py > gpurun_out/r4s/ab_wwaves.log 2>&1; cat gpurun_out/r4s/ab_wwaves.log
CHECK=1 TUNE=9=3 KNOB=20 VALUES=0,8,12,16 NETS=14 FRAMES=49 ROUNDS=2 timeout 400 python scripts/ab_knob_train.py > gpurun_out/r4s/ab_wwaves14.log 2>&1; cat gpurun_out/r4s/ab_wwaves14.log
