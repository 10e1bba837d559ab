#!/bin/bash
# spread of the headline over repeated runs on ONE box (driver's K / W), and at a longer K
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-eval --no-epoch --no-lightgcn --no-gru --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=20 run $i value=%.0f ms/step=%.4f cold_ms=%.3f'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
python bench.py --gpus 1 --steps 200 --warmup 20 --no-eval --no-epoch --no-lightgcn --no-gru --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=200 value=%.0f ms/step=%.4f'%(d['value'], d['ms_per_step']))"
python -m pytest tests/test_gpu_gru.py tests/test_gpu_sampler.py -x -q 2>&1 | tail -2
