#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_pool
mkdir -p $O
cd $R
for lib in $R/rsoderh-raytracing_amd/librsrt.so $R/rsoderh-raytracing_amd/librsrt_exp_463608f683.so; do
  RSRT_LIB=$lib timeout -k 10 200 python tools/bvh_knobs.py "RSRT_TRAVERSAL=4;RSRT_TRAVERSAL=4,RSRT_STOP_QUORUM=60" 2>&1 | grep -v "round 0"
done > $O/pool2.txt 2>&1
cat $O/pool2.txt
