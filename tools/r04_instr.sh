#!/bin/bash
# instrumented build: stage shares and loop-trip lane efficiency of the general-BVH configs under the current knobs
O=gpurun_out/r04_instr; mkdir -p $O; rm -f $O/instr.txt
GRID=$(python -c "import sys; sys.path.insert(0,'tools'); import make_big_scene; print(make_big_scene.make(4))")
timeout -k 10 200 python tools/simd_efficiency.py 4 32 suzanne 1280 720 10 >> $O/instr.txt 2>&1 &&
timeout -k 10 200 python tools/simd_efficiency.py 4 8 $GRID 1280 720 10 >> $O/instr.txt 2>&1 || { tail -20 $O/instr.txt; exit 1; }
grep -v "descend\|leaf loop\|per TRACE inv\|outer trips\|SHADE: NEE" $O/instr.txt
