#!/bin/bash
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q -k "cold_pass or blocked or fused" > gpurun_out/r2_cold_tests.log 2>&1 || { tail -30 gpurun_out/r2_cold_tests.log; exit 1; }
tail -2 gpurun_out/r2_cold_tests.log
for K in 20 200; do
  timeout -k 10 300 python bench.py --gpus 1 --steps $K --warmup 5 --no-lightgcn --no-gru --no-epoch --no-cpu-baseline --no-eval > gpurun_out/r2_cold_bench.json 2> gpurun_out/r2_cold_bench.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2_cold_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("K=$K", "value %.3g" % d["value"], "ms/step %.4f" % d["ms_per_step"], "roofline", round(r["frac"],3), round(r["avg_launch_ms"],3), "alone", r.get("alone",{}).get("avg_launch_ms"), r.get("alone",{}).get("frac"))
PY
done
