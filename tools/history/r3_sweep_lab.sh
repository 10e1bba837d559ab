#!/bin/bash
cd $GRAFT_REPO_ROOT
make -C tools/lab libsweep_lab.so > /dev/null 2>&1
timeout -k 10 900 python3 tools/sweep_lab.py > gpurun_out/r3_sweep_lab.txt 2> gpurun_out/r3_sweep_lab.err; echo "exit $?"
cat gpurun_out/r3_sweep_lab.txt; tail -5 gpurun_out/r3_sweep_lab.err
