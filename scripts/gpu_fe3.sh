#!/bin/bash
# one gpurun call: front-end parity tests, the three-waves vs two-waves A/B (with sweeps), a quick headline bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/fe3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "frontend" > gpurun_out/fe3/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/fe3/pytest.log
timeout 600 python scripts/ab_fe_kernel.py --sweep > gpurun_out/fe3/ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/fe3/ab.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs ${LEGS:-train} > gpurun_out/fe3/bench.json 2> gpurun_out/fe3/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
o = json.loads(open("gpurun_out/fe3/bench.json").read().strip().splitlines()[-1])
print(o["ms_per_step"], o["value"], "seq", o["sequential"]["ms_per_step"], "pair", o["pair_roofline"]["frac"], "fe", o["roofline"]["frac"], o["roofline"]["kernel"], o["roofline"]["kernel_ms"], o.get("bitwise_equal_to_sequential"),
      {k: o[k]["ms_per_step"] for k in o if isinstance(o[k], dict) and "ms_per_step" in o[k] and k != "sequential"})
PY
