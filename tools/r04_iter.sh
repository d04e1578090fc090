#!/bin/bash
# cooperative walk, one iteration of work: parity subset, timing of the two general-BVH configs (+ house), instrumented run
O=gpurun_out/r04_iter; mkdir -p $O; rm -f $O/*.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cooperative or ray_batch or kernel_variant or mid_size or big_scene or chain_tree or deep_tree or twin_records or coincident" > $O/tests.txt 2>&1
rc=$?; tail -3 $O/tests.txt
if [ $rc -ne 0 ]; then echo "tests failed ($rc): no timing"; exit $rc; fi
for i in 1 2; do
timeout -k 10 200 python tools/scene_time.py suzanne 1280 720 128 10 >> $O/ab.txt 2>&1 &&
timeout -k 10 200 python tools/scene_time.py grid4 1280 720 32 10 >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
done
grep -v "^$" $O/ab.txt | awk 'NR%3==0'
bash tools/r04_instr.sh | grep "cooperative\|popped\|wave cycles\|wave-time"
