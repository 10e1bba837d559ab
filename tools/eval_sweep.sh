# usage: bash tools/eval_sweep.sh "<label>"  -- fused-eval throughput at several launch sizes, v1 vs v2
for v in 1 2; do for eu in 65536 131072 262144; do
  SKR_FUSED_V=$v python bench.py --steps 5 --warmup 1 --no-cpu-baseline --eval-users $eu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v$v users=$eu', 'TF=%.1f'%d['roofline_eval']['achieved'], 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'], 'users/s=%.0f'%d['eval']['users_per_sec'], 'HR=%.5f'%d['eval']['HR@10'], 'NDCG=%.6f'%d['eval']['NDCG@10'])"
done; done
