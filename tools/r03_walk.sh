#!/bin/bash
# the wide walk: parity, timings against the fixed-order walk, loop-trip counters   (bash tools/r03_walk.sh <tag> "<knob matrix>")
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 100 -k "ray_batch or big_scene or mid_size or every_kernel or twin or coincident or axis_parallel or random_scenes or matches_oracle_live" > $O/wide_tests.txt 2>&1; echo "pytest exit $?" >> $O/wide_tests.txt
tail -4 $O/wide_tests.txt
grep -q "pytest exit 0" $O/wide_tests.txt || exit 1
timeout -k 10 400 python tools/bvh_knobs.py "$2" > $O/knobs.txt 2>&1; cat $O/knobs.txt
python -c "import sys; sys.path.insert(0,'tools'); import make_big_scene; make_big_scene.make(4)"
RSRT_TRAVERSAL=4 timeout -k 10 200 python tools/simd_efficiency.py 4 16 /tmp/rsrt_scenes/suzanne_grid_4.toml 1280 720 10 > $O/simd_grid_t4.txt 2>&1; cat $O/simd_grid_t4.txt
RSRT_TRAVERSAL=4 timeout -k 10 200 python tools/simd_efficiency.py 4 64 suzanne 1280 720 10 > $O/simd_suzanne_t4.txt 2>&1; cat $O/simd_suzanne_t4.txt
