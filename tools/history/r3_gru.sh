#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_gru.py tests/test_gpu_train.py -q -x -k "gru or session or blocked or fit or config4 or pop" > gpurun_out/r3_gru_tests.log 2>&1; echo "tests exit $?"; tail -6 gpurun_out/r3_gru_tests.log
timeout -k 10 300 python3 bench.py --workload gru4rec --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3_gru.json 2> gpurun_out/r3_gru.err; echo "gru exit $?"; tail -3 gpurun_out/r3_gru.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_gru.json').read().strip().splitlines()[-1])
print('gru ms/step', d['ms_per_step'], d['value'], json.dumps(d['roofline'])[:600])"
SKR_ADAM_BLOCK=1 timeout -k 10 300 python3 bench.py --workload gru4rec --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3_gru_k1.json 2> gpurun_out/r3_gru_k1.err; echo "gru k1 exit $?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_gru_k1.json').read().strip().splitlines()[-1])
print('gru k=1 ms/step', d['ms_per_step'], d['value'])"
