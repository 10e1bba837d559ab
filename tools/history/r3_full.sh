#!/bin/bash
# the whole GPU suite + smoke + the driver's bench line
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3_full_tests.log 2>&1; echo "tests exit $?"; tail -8 gpurun_out/r3_full_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/r3_smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err; echo "bench exit $?"
tail -c 300 gpurun_out/r3_bench_n1.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_bench_n1.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "epochs", d.get("full_epoch", {}).get("epochs_seconds"))
for k in ("roofline", "roofline_fresh_model"):
    r = d.get(k, {}); print(k, r.get("frac"), r.get("avg_launch_ms"), r.get("traffic"))
r = d["roofline_step"]; print("step", r["avg_launch_us"], r["alone"]["avg_launch_us"], r["traffic"])
print("large", {k: (v["value"], v["form"]) for k, v in d.get("large_batch", {}).items()})
print("eval", d.get("eval"), d.get("roofline_eval", {}).get("frac"))
lg = d.get("lightgcn", {}); print("lightgcn", lg.get("ms_per_step"), lg.get("roofline", {}).get("avg_launch_ms"), {k: v["ms_per_step"] for k, v in lg.get("large_batch", {}).items()})
g = d.get("gru4rec", {}); print("gru", g.get("ms_per_step"), g.get("value"))
PY
