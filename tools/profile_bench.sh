#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats            per-kernel time
#   2. --pmc FETCH_SIZE   (own pass)     HBM read traffic  (KB; x2 for wide coalesced streams on gfx950)
#   3. --pmc WRITE_SIZE   (own pass)     HBM write traffic (KB)
#   4. --pmc TCC_HIT_sum TCC_MISS_sum    (own pass) L2 hit rate of the propagation kernels
#   5. --pmc TCP_TCC_READ_REQ_sum        (own pass) read requests from the CUs' vector caches to L2
# PMC passes never carry --stats / sys-trace flags.  Outputs land under gpurun_out/prof_<tag>/.
# The command is the DRIVER's (python3 bench.py --gpus 1 --steps 20 --warmup 5) without the host-side legs: the CPU
# baseline and the three whole epochs (3 x 47 k steps in the trace) change no kernel of the timed region or the LightGCN leg.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-epoch"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.stats.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG -o fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG -o write -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_$TAG -o l2 -- python3 $R/bench.py $ARGS --no-eval > $R/gpurun_out/prof_$TAG.l2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/prof_$TAG -o tcp -- python3 $R/bench.py $ARGS --no-eval > $R/gpurun_out/prof_$TAG.tcp.log 2>&1
echo "profile exit $?"
ls -la $R/gpurun_out/prof_$TAG
cd $R && python3 tools/summarize_profiles.py $TAG > gpurun_out/prof_$TAG.summary.log 2>&1; echo "summary exit $?"
