# usage: bash tools/bench_2rank.sh -- rehearsal of bench.py's N = 2 path on ONE GPU (gloo collectives; timing is
# not meaningful for xGMI, correctness of the orchestration is): both exchange modes
for ex in sparse dense; do
  SKR_EXCHANGE=$ex SKR_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline --eval-users 65536 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ex', 'value=%.0f'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], 'HR=%s'%d.get('eval',{}).get('HR@10'))"
done
