#!/bin/bash
# small-batch latency: the small-batch network kernel (default) against the throughput kernel at one utterance per group (knob 27 = 1)
cd $GRAFT_REPO_ROOT
for B in 1 8 64; do for K in 0 1; do echo "knob27=$K"; B=$B TUNE=27=$K STEPS=300 timeout 120 python scripts/latency_b1.py 2>&1 | grep "per call + sync"; done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small or batch_indep or waveform or every_kernel or eval_forward or full_batch_eval" 2>&1 | tail -3
mkdir -p gpurun_out/small; cd /tmp; export TMPDIR=/tmp
B=1 STEPS=50 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/small/trace -o t -- python $GRAFT_REPO_ROOT/scripts/latency_b1.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/small/trace/**/*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"][:70], r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3)
PY
