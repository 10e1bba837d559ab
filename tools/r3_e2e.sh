#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3_e2e_api.txt
echo "# tools/e2e_scale.py on one MI355X (gpurun), drop-in API end to end, round 3 (TSV in the reference's format -> RunConfig -> Model(...) -> train_epoch / evaluate)" > $OUT
echo "# BPRMF, 1 000 000 users / 100 000 items / ~48.4 M train interactions, batch 1024, exact sampler, 4 epochs" >> $OUT
timeout -k 10 700 python3 tools/e2e_scale.py --users 1000000 --items 100000 --interactions 50000000 --epochs 4 2>/dev/null | grep "\[e2e\]" >> $OUT; echo "bprmf exit $?"
echo "# LightGCN 3 layers, same data, 100 steps (E2E_MAX_STEPS=100; every step is the full-graph propagation forward and backward)" >> $OUT
E2E_MAX_STEPS=100 timeout -k 10 400 python3 tools/e2e_scale.py --users 1000000 --items 100000 --interactions 50000000 --epochs 2 --model LightGCN 2>/dev/null | grep "\[e2e\]" >> $OUT; echo "lightgcn exit $?"
echo "# BPRMF n_dim=128 (rows of 128 floats: skr_bpr_step_dim + a dense Adam launch per step; score-matrix evaluation), 200 000 users / 20 000 items" >> $OUT
cat $OUT
