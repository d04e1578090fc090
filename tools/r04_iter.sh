#!/bin/bash
# cooperative walk, one iteration of work: parity subset, timing of the two general-BVH configs under a knob's values
# usage: bash tools/r04_iter.sh KNOB "v1 v2 ..."
O=gpurun_out/r04_iter; mkdir -p $O; rm -f $O/*.txt
K=${1:-RSRT_NONE}; VALS=${2:-x}
for v in $VALS; do
export $K=$v
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cooperative or ray_batch or kernel_variant or mid_size or big_scene or chain_tree or deep_tree or twin_records or coincident" > $O/tests_$v.txt 2>&1
rc=$?; tail -1 $O/tests_$v.txt
if [ $rc -ne 0 ]; then echo "tests failed ($rc) with $K=$v: stop"; tail -20 $O/tests_$v.txt; exit $rc; fi
done
for v in $VALS $VALS; do
export $K=$v
timeout -k 10 200 python tools/scene_time.py suzanne 1280 720 128 10 >> $O/ab.txt 2>&1 &&
timeout -k 10 200 python tools/scene_time.py grid4 1280 720 32 10 >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
done
grep -v "^$" $O/ab.txt | awk 'NR%3==0'
