#!/bin/bash
# Round 4, first GPU call for the cooperative wide walk (TRAV 6): its parity tests, then an A/B against the one-ray-a-lane wide walk (TRAV 4)
# on the two general-BVH configs.  Run through gpurun from the repo root; everything lands in gpurun_out/r04_walk/.
O=gpurun_out/r04_walk; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cooperative or ray_batch or probe_refuses or kernel_variant or mid_size or big_scene or chain_tree or deep_tree or twin_records or coincident or axis_parallel" > $O/tests.txt 2>&1
rc=$?; tail -5 $O/tests.txt
if [ $rc -ne 0 ]; then echo "tests failed ($rc): no timing"; exit $rc; fi
for t in 6 4; do
  RSRT_TRAVERSAL=$t timeout -k 10 200 python tools/scene_time.py suzanne 1280 720 128 10 >> $O/ab.txt 2>&1 &&
  RSRT_TRAVERSAL=$t timeout -k 10 200 python tools/scene_time.py grid4 1280 720 32 10 >> $O/ab.txt 2>&1 || exit 1
done
cat $O/ab.txt
