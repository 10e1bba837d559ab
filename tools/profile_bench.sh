#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats            per-kernel time
#   2. --pmc FETCH_SIZE   (own pass)     HBM read traffic  (KB; x2 for wide coalesced streams on gfx950)
#   3. --pmc WRITE_SIZE   (own pass)     HBM write traffic (KB)
# PMC passes never carry --stats / sys-trace flags.  Outputs land under gpurun_out/prof_<tag>/.
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
ARGS="--steps 192 --warmup 32 --no-cpu-baseline --no-epoch"   # multiples of the Adam block (32); the whole-epoch leg would add 3 x 47 k steps to the trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.stats.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG -o fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG -o write -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.write.log 2>&1
echo "profile exit $?"
ls -la $R/gpurun_out/prof_$TAG
