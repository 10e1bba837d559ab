# usage: bash tools/eval_sweep.sh  -- fused-eval throughput at several launch sizes (env passes through)
for eu in 4096 65536 131072 262144; do
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --eval-users $eu 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V=%s ABL=%s TRIG=%s users=$eu'%(os.environ.get('SKR_FUSED_V','2'),os.environ.get('SKR_FUSED_ABLATE','0'),os.environ.get('SKR_FUSED_TRIGGER','-')), 'TF=%.1f'%d['roofline_eval']['achieved'], 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'], 'users/s=%.0f'%d['eval']['users_per_sec'], 'HR=%.5f'%d['eval']['HR@10'], 'NDCG=%.6f'%d['eval']['NDCG@10'])"
done
