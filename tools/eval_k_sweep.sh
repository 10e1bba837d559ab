# usage: bash tools/eval_k_sweep.sh -- fused-eval throughput vs top_k, list capacity and compaction trigger
for k in 10 20 50 100; do for cap in 256 512; do for tr in 0 $((k+48)) $((cap-32)); do
  SKR_FUSED_CAP=$cap SKR_FUSED_TRIGGER=$tr python bench.py --steps 5 --warmup 1 --no-cpu-baseline --eval-users 262144 --top-k $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('top_k=$k cap=$cap trigger=$tr', 'TF=%.1f'%d['roofline_eval']['achieved'], 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'], 'NDCG=%.6f'%d['eval'].get('NDCG@$k', d['eval'].get('NDCG@10', 0)))"
done; done; done
