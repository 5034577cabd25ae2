#!/bin/bash
# The round's closing GPU call: smoke + every GPU test + bench (defaults, then the driver's flags) + the three training-step profile packs.
#   gpurun --timeout 1800 -- 'bash scripts/gpu_final_round.sh r06z'
TAG=${1:-final}
cd $GRAFT_REPO_ROOT
bash scripts/gpu_round.sh $TAG
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench_driver_flags.json 2> gpurun_out/$TAG/bench_driver_flags.err; echo "bench (driver flags) rc=$?"
bash scripts/gpu_prof_train_all.sh
