#!/bin/bash
# PMC calibration of the cold pass's 4-byte-per-lane accesses (MI355X_MICROARCH.md, HBM: FETCH_SIZE / WRITE_SIZE are
# calibrated for 16-byte-per-lane streams only): tools/microbench_cold.py in states with a KNOWN byte count per launch
#   lively  every block on the ordinary path: 12 B read + 12 B written per parameter
#   old     every block at rest:              12 B read +  8 B written per parameter (p is not written back)
# Separate --pmc passes, no --stats / sys-trace beside them.  Output: gpurun_out/prof_<tag>_cold/
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for st in lively old; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_cold -o fetch_$st -- python3 $R/tools/microbench_cold.py --states $st --ks 24 --t0s 20000 > $R/gpurun_out/prof_${TAG}_cold_$st.fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_cold -o write_$st -- python3 $R/tools/microbench_cold.py --states $st --ks 24 --t0s 20000 > $R/gpurun_out/prof_${TAG}_cold_$st.write.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, json
out = {}
for st in ("lively", "old"):
    for name in ("fetch", "write"):
        f = glob.glob("$R/gpurun_out/prof_${TAG}_cold/**/%s_%s_counter_collection.csv" % (name, st), recursive=True)[0]
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "adam_cold_rows_kernel" in r["Kernel_Name"]]
        out["%s_%s_KB" % (st, name)] = sum(v) / len(v)
n = 70500000
out["lively_read_factor"] = 12.0 * n / (out["lively_fetch_KB"] * 1024)
out["lively_write_factor"] = 12.0 * n / (out["lively_write_KB"] * 1024)
out["old_read_factor"] = 12.0 * n / (out["old_fetch_KB"] * 1024)
out["old_write_factor"] = 8.0 * n / (out["old_write_KB"] * 1024)
print(json.dumps(out))
open("$R/gpurun_out/prof_${TAG}_cold/calibration.json", "w").write(json.dumps(out, indent=1))
PY
