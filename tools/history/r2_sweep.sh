#!/bin/bash
# knobs around the fused step and the cold pass: the driver's command (K = 20) and whole epochs
run() {
  env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-lightgcn --no-gru --no-cpu-baseline --no-eval > gpurun_out/r2_sweep.json 2> gpurun_out/r2_sweep.err || { tail -5 gpurun_out/r2_sweep.err; return 1; }
  python - "$*" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r2_sweep.json").read().strip().splitlines()[-1])
r=d["roofline"]; e=d["full_epoch"]
print(sys.argv[1].ljust(44), "value %.4g" % d["value"], "cold %.3f ms frac %.3f" % (r["avg_launch_ms"], r["frac"]), "epoch %.4g /s %.3f s (host queued %.3f) second %.3f first %.3f" % (e["interactions_per_sec"], e["seconds"], e["host_queued_after_seconds"], e["epoch_drawing_ahead_too_seconds"], e["first_epoch_seconds"]))
PY
}
for cfg in "$@"; do run $cfg || exit 1; done
