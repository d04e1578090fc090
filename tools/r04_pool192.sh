#!/bin/bash
# 12 waves x 192 slots (768-thread workgroup) against the product's 16 waves x 128 slots for the cooperative walk: parity subset first
O=gpurun_out/r04_pool192; mkdir -p $O; rm -f $O/*.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/rsoderh-raytracing_amd
for lib in $P/librsrt_exp_03c2363e96.so; do
RSRT_LIB=$lib timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cooperative or ray_batch or kernel_variant or mid_size or big_scene or chain_tree or deep_tree or twin_records or coincident" > $O/tests.txt 2>&1
rc=$?; tail -1 $O/tests.txt
if [ $rc -ne 0 ]; then echo "tests failed ($rc): stop"; tail -30 $O/tests.txt; exit $rc; fi
done
for pass in 1 2; do
  for v in product w12_p192_wps3 w12_p192_wps4; do
    unset RSRT_LIB
    [ $v = w12_p192_wps3 ] && export RSRT_LIB=$P/librsrt_exp_03c2363e96.so
    [ $v = w12_p192_wps4 ] && export RSRT_LIB=$P/librsrt_exp_c284636d86.so
    echo "== $v" >> $O/ab.txt
    timeout -k 10 200 python tools/scene_time.py suzanne 1280 720 128 10 2>&1 | tail -1 >> $O/ab.txt &&
    timeout -k 10 200 python tools/scene_time.py grid4 1280 720 32 10 2>&1 | tail -1 >> $O/ab.txt || { tail $O/ab.txt; exit 1; }
  done
done
sed 's/RSRT_LIB=[^ ]*//' $O/ab.txt | cut -c1-120
