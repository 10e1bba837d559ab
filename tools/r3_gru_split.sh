#!/bin/bash
# GRU4RecPlus leg with the k-split forward kernel (1) or gru_fwd_kernel (0), same box; then the GRU tests
cd $GRAFT_REPO_ROOT
for v in ${@:-1 0 1 0}; do
  SKR_GRU_SPLIT=$v timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --gru-steps 200 --no-cpu-baseline --no-eval --no-epoch --no-lightgcn --large-batches "" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
g = d['gru4rec']
print('split $v ms/step', round(g['ms_per_step'], 4), 'events/s', round(g['value']))" || exit 1
done
timeout -k 10 600 python -m pytest tests/test_gpu_gru.py -x -q -m gpu 2>&1 | tail -3
