#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/rsoderh-raytracing_amd
timeout -k 10 300 python $R/tools/occ_sweep.py "2:4" 32
cp $P/librsrt.so $P/librsrt_w4.so; cp $P/librsrt_w5.so $P/librsrt.so
timeout -k 10 300 python $R/tools/ab_kernels.py 2 32 2>&1 | grep -E "False|round 2"
timeout -k 10 300 python $R/tools/occ_sweep.py "2:4,5" 32
