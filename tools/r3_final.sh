#!/bin/bash
# final check of the round: whole GPU suite, smoke, the driver's bench line, the self-launching N = 2 rehearsal (gloo, one card)
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3b_final_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/r3b_final_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3b_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r3b_smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3b_bench_n1.json 2> gpurun_out/r3b_bench_n1.err; echo "bench exit $?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3b_bench_n1.json").read().strip().splitlines()[-1])
print("value", d["value"], d["repeats"]["seconds"], "epochs", d.get("full_epoch", {}).get("epochs_seconds"))
r = d["roofline"]; print("roofline", r["frac"], r["avg_launch_ms"], r["traffic"], r.get("frac_by_traffic"))
r = d["roofline_step"]; print("step", r["avg_launch_us"], r["alone"]["avg_launch_us"], r["traffic"])
print("large", {k: (v["value"], v["form"]) for k, v in d.get("large_batch", {}).items()})
print("eval", d.get("eval", {}).get("users_per_sec"), d.get("roofline_eval", {}).get("frac"))
lg = d.get("lightgcn", {}); print("lightgcn", lg.get("ms_per_step"), lg.get("roofline", {}).get("avg_launch_ms"), {k: v["ms_per_step"] for k, v in lg.get("large_batch", {}).items()}, "layergcn", lg.get("layergcn", {}).get("ms_per_step"))
g = d.get("gru4rec", {}); print("gru", g.get("ms_per_step"), g.get("value"))
PY
env -u WORLD_SIZE -u RANK -u LOCAL_RANK SKR_DIST_BACKEND=gloo timeout -k 10 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-epoch > gpurun_out/r3b_bench_n2_gloo.json 2> gpurun_out/r3b_bench_n2_gloo.err; echo "n2 exit $?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3b_bench_n2_gloo.json").read().strip().splitlines()[-1])
print("n2:", d["n_gpus"], d["rccl_ranks"], d["dist_backend"], d["config"].get("item_table_replicas_identical"), d["lightgcn"]["ms_per_step"], d["gru4rec"]["ms_per_step"])
PY
