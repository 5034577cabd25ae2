#!/bin/bash
# one gpurun call: bench.py with the driver's flags (headline + the legs in $LEGS), no CPU baseline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/quick
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs ${LEGS:-train} > gpurun_out/quick/bench_driver.json 2> gpurun_out/quick/bench_driver.err; echo "rc=$?"
python - <<'PY'
import json
o = json.loads(open("gpurun_out/quick/bench_driver.json").read().strip().splitlines()[-1])
print(o["ms_per_step"], o["value"], "seq", o["sequential"]["ms_per_step"], "pair", o["pair_roofline"]["frac"], "fe", o["roofline"]["frac"], o["roofline"]["kernel"], o["roofline"]["kernel_ms"], o.get("bitwise_equal_to_sequential"),
      {k: o[k]["ms_per_step"] for k in o if isinstance(o[k], dict) and "ms_per_step" in o[k] and k != "sequential"}, o.get("latency_batch_1", {}).get("value"))
PY
