#!/bin/bash
# round 3, first GPU call: the suite, the driver's bench line, the self-launching N = 2 rehearsal (gloo, one card)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/r3_tests.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err; echo "bench exit $?"
tail -c 600 gpurun_out/r3_bench_n1.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3_bench_n1.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "repeats", d["repeats"]["seconds"])
    print("epoch", d.get("full_epoch", {}).get("epochs_seconds"))
    for k in ("roofline", "roofline_fresh_model"):
        r = d.get(k, {})
        print(k, r.get("frac"), r.get("avg_launch_ms"), r.get("launches_averaged"), r.get("hot_blocks"))
    print("step", json.dumps(d.get("roofline_step"))[:900])
    print("large", json.dumps(d.get("large_batch"))[:2500])
    print("eval", d.get("eval"), d.get("roofline_eval", {}).get("frac"))
    lg = d.get("lightgcn", {})
    print("lightgcn", lg.get("ms_per_step"), lg.get("roofline", {}).get("avg_launch_ms"), lg.get("large_batch"))
    print("gru", d.get("gru4rec", {}).get("ms_per_step"))
except Exception as e:
    print("parse failed", e)
PY
env -u WORLD_SIZE -u RANK -u LOCAL_RANK SKR_DIST_BACKEND=gloo timeout -k 10 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-epoch > gpurun_out/r3_bench_n2_gloo.json 2> gpurun_out/r3_bench_n2_gloo.err; echo "n2 exit $?"
tail -c 400 gpurun_out/r3_bench_n2_gloo.err
head -c 700 gpurun_out/r3_bench_n2_gloo.json
