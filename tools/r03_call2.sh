#!/bin/bash
# round 3, GPU call 2: the wide walk — parity first (own timeout), then timings against the fixed-order walk
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_call2
mkdir -p $O
cd $R
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 100 -k "ray_batch or big_scene or mid_size or every_kernel or twin or coincident or axis_parallel" > $O/wide_tests.txt 2>&1; echo "pytest exit $?" >> $O/wide_tests.txt
tail -15 $O/wide_tests.txt
grep -q "pytest exit 0" $O/wide_tests.txt || exit 1
timeout -k 10 500 python tools/bvh_knobs.py "RSRT_TRAVERSAL=3;RSRT_TRAVERSAL=4;RSRT_TRAVERSAL=4,RSRT_TRACE_BUDGET=2;RSRT_TRAVERSAL=4,RSRT_TRACE_BUDGET=8;RSRT_TRAVERSAL=4,RSRT_STOP_QUORUM=0,RSRT_TRACE_BUDGET=64;RSRT_TRAVERSAL=4,RSRT_STOP_QUORUM=60;RSRT_TRAVERSAL=4,RSRT_DESCEND_QUORUM=50;RSRT_TRAVERSAL=4,RSRT_DESCEND_QUORUM=15;RSRT_TRAVERSAL=4,RSRT_HYBRID=0" > $O/knobs.txt 2>&1; cat $O/knobs.txt
