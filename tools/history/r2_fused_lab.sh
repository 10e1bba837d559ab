#!/bin/bash
for d in 0 1 2 4 8 3 7 15; do SKR_FUSED_DBG=$d timeout -k 10 120 python tools/fused_lab.py 32 10 || exit 1; done

SKR_FUSED_DBG=0 timeout -k 10 120 python tools/fused_lab.py 8 40
SKR_FUSED_DBG=0 timeout -k 10 120 python tools/fused_lab.py 64 6
