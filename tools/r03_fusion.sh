#!/bin/bash
# round 3: bounded experiments on the house kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_fusion
mkdir -p $O
cd $R
timeout -k 10 300 python tools/house_knobs.py "RSRT_CULL_DEPTH=2;RSRT_CULL_DEPTH=1;RSRT_CULL_DEPTH=3" 64 > $O/depth.txt 2>&1; cat $O/depth.txt
timeout -k 10 500 bash tools/lib_ab.sh 64 > $O/ab2.txt 2>&1; cat $O/ab2.txt
