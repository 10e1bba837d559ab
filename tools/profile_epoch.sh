#!/bin/bash
# rocprofv3 passes over the WHOLE-EPOCH state of bench.py (round 3): the cold pass and the one-launch step as an epoch runs
# them (every row's moments aged by real training) -- what `roofline` / `roofline_step` of the bench line report.
#   1. --kernel-trace --stats                      per-kernel time over pre-steps, timed steps and three whole epochs
#   2. --pmc FETCH_SIZE  (own pass, kernel filter)  HBM read traffic of adam_cold_rows_kernel / bpr_fused_step_kernel / _end
#   3. --pmc WRITE_SIZE  (own pass, kernel filter)  HBM write traffic
# PMC passes never carry --stats / sys-trace flags.  The counter CSVs hold one row per dispatch (150 k launches per pass):
# they are condensed on the box (tools/summarize_epoch_profile.py) and deleted; only the summary travels back.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
ARGS="--gpus 1 --steps 20 --warmup 5 --no-lightgcn --no-gru --no-cpu-baseline --no-eval --large-batches="
OUT=$R/gpurun_out/prof_epoch_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 $R/bench.py $ARGS > $OUT.stats.json 2> $OUT.stats.err &&
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "adam_cold_rows_kernel|bpr_fused_step_kernel|bpr_fused_end_kernel|bpr_fused_pre_kernel" --output-format csv -d $OUT -o fetch -- python3 $R/bench.py $ARGS > $OUT.fetch.json 2> $OUT.fetch.err &&
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "adam_cold_rows_kernel|bpr_fused_step_kernel|bpr_fused_end_kernel|bpr_fused_pre_kernel" --output-format csv -d $OUT -o write -- python3 $R/bench.py $ARGS > $OUT.write.json 2> $OUT.write.err
echo "profile exit $?"
cd $R && python3 tools/summarize_epoch_profile.py $TAG > $OUT.summary.log 2>&1; echo "summary exit $?"; tail -40 $OUT.summary.log
rm -f $OUT/*_kernel_trace.csv $OUT/*_counter_collection.csv
ls -la $OUT
