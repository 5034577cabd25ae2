#!/bin/bash
# one gpurun call: the pipeline parity tests + bench.py with the driver's flags and with its defaults (headline only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/quick
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipeline" > gpurun_out/quick/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/quick/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs ${LEGS:-none} > gpurun_out/quick/bench_driver.json 2> gpurun_out/quick/bench_driver.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/quick/bench_driver.json",):
    o = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, o["ms_per_step"], o["value"], "seq", o["sequential"]["ms_per_step"], "pair", o["pair_roofline"]["frac"], "fe", o["roofline"]["frac"], o.get("bitwise_equal_to_sequential"),
          {k: o[k]["ms_per_step"] for k in o if isinstance(o[k], dict) and "ms_per_step" in o[k] and k != "sequential"})
PY
timeout 300 python bench.py --no-cpu-baseline --legs ${LEGS:-none} > gpurun_out/quick/bench_default.json 2> gpurun_out/quick/bench_default.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/quick/bench_default.json",):
    o = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, o["ms_per_step"], o["value"], "seq", o["sequential"]["ms_per_step"], "pair", o["pair_roofline"]["frac"], "fe", o["roofline"]["frac"], o.get("bitwise_equal_to_sequential"),
          {k: o[k]["ms_per_step"] for k in o if isinstance(o[k], dict) and "ms_per_step" in o[k] and k != "sequential"})
PY
