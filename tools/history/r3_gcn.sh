#!/bin/bash
# round 3: the propagation's row epilogues -- touched tests, then the LightGCN / LayerGCN steps at full size
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_models.py tests/test_gpu_dist.py tests/test_gpu_config0.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -q -k "spmm or lightgcn or layergcn or graph or refine or sharded or torchrun or rccl or bench_contract or propagation" > gpurun_out/r3_gcn_tests.log 2>&1; echo "tests exit $?"; tail -8 gpurun_out/r3_gcn_tests.log
timeout -k 10 300 python3 tools/layergcn_fullsize.py 10 > gpurun_out/r3_layergcn.txt 2>&1; echo "layergcn exit $?"; tail -2 gpurun_out/r3_layergcn.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lgcn -o lg -- python3 $R/bench.py --workload lightgcn --steps 10 --warmup 2 --no-cpu-baseline --large-batches "" > $R/gpurun_out/r3_lightgcn.json 2> $R/gpurun_out/r3_lightgcn.err; echo "lightgcn exit $?"
cd $R
python3 - <<'PY'
import csv, json
d=json.loads(open('gpurun_out/r3_lightgcn.json').read().strip().splitlines()[-1])
print('lightgcn ms/step', d['ms_per_step'], 'layer', d['roofline']['avg_launch_ms'], d['roofline']['user_side_ms'], d['roofline']['item_side_ms'])
rows=list(csv.DictReader(open("gpurun_out/prof_lgcn/lg_kernel_stats.csv")))
for r in rows[:25]:
    print(r["Name"][:70].ljust(70), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
rm -f gpurun_out/prof_lgcn/lg_kernel_trace.csv
