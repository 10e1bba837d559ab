#!/bin/bash
# round-2 GPU check: touched tests, the driver's default bench command, 2-rank rehearsals (gloo, one GPU) of both scaling modes
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gru.py tests/test_gpu_train.py tests/test_gpu_models.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -x -q -k "gru or session or pop or spmm or lightgcn or layergcn or dist" > gpurun_out/r2_tests4.log 2>&1; echo "tests rc=$?"
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_bench_n1.json 2> gpurun_out/r2_bench_n1.err; echo "bench rc=$?"
# (two processes time-slicing ONE GPU: side streams only add cross-queue waits there, so the rehearsal keeps everything on
#  one stream per process; on a real node every rank has its GPU to itself)
for mode in strong weak; do
  SKR_ADAM_OVERLAP=0 SKR_SAMPLER_ONE_STREAM=1 SKR_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline --eval-users 65536 \
    --users 200000 --items 20000 --interactions 10000000 --pre-steps 256 --lightgcn-steps 4 --scaling $mode \
    > gpurun_out/r2_bench_2rank_$mode.json 2> gpurun_out/r2_bench_2rank_$mode.err; echo "2rank $mode rc=$?"
done
