# usage: bash tools/eval_ablate_sweep.sh -- fused-eval cost split at large top_k: SKR_FUSED_ABLATE 5 runs every
# compaction's sort twice, 6 the train masking twice; the added time is that phase's cost
for k in 10 50 100; do for ab in 0 5 6; do
  SKR_FUSED_ABLATE=$ab python bench.py --steps 5 --warmup 1 --no-cpu-baseline --eval-users 262144 --top-k $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('top_k=$k ablate=$ab', 'TF=%.1f'%d['roofline_eval']['achieved'], 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'])"
done; done
