# usage: bash tools/bench_ranks.sh <N> [bench args] -- rehearsal of bench.py on N gloo ranks sharing ONE GPU
# (correctness of the multi-rank orchestration only; timings say nothing about xGMI).  N <= 6 on a gpurun box.
N=${1:-2}; shift
SKR_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
  --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $N --no-cpu-baseline "$@" 2>&1 | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('n_gpus', d['n_gpus'], 'value=%.0f'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], 'replicas identical:', d['config'].get('item_table_replicas_identical'), 'roofline kernel:', d['roofline']['kernel'][:40], 'eval:', d.get('eval'))"
