#!/bin/bash
# leaf trips at record granularity (rt_coop.h): parity subset with the new build, then the two general-BVH configs timed with the previous build
# (librsrt_exp_prev.so: one leaf item a lane, records in pairs), the previous build with its node ring capped at the new build's 320 entries, and the new one
O=gpurun_out/r04_leaf; mkdir -p $O; rm -f $O/*.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cooperative or ray_batch or kernel_variant or mid_size or big_scene or chain_tree or deep_tree or twin_records or coincident" > $O/tests.txt 2>&1
rc=$?; tail -1 $O/tests.txt
if [ $rc -ne 0 ]; then echo "tests failed ($rc): stop"; tail -30 $O/tests.txt; exit $rc; fi
PREV=$R/rsoderh-raytracing_amd/librsrt_exp_prev.so
for pass in 1 2; do
  for v in prev new; do
    unset RSRT_LIB RSRT_COOP_LDS_CAP
    [ $v = prev ] && export RSRT_LIB=$PREV
    [ $v = prev320 ] && export RSRT_LIB=$PREV RSRT_COOP_LDS_CAP=320
    echo "== $v" >> $O/ab.txt
    timeout -k 10 200 python tools/scene_time.py suzanne 1280 720 128 10 >> $O/ab.txt 2>&1 &&
    timeout -k 10 200 python tools/scene_time.py grid4 1280 720 32 10 >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
  done
done
grep -v "^$" $O/ab.txt | grep "==\|trace\|ms" | cut -c1-200
