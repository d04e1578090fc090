#!/bin/bash
# cooperative walk: how long TRACE waits for the other stages (RSRT_COOP_DEFER), timing + one instrumented run
O=gpurun_out/r04_defer; mkdir -p $O
for d in 0 1 24 40 56 65; do
  RSRT_COOP_DEFER=$d timeout -k 10 200 python tools/scene_time.py suzanne 1280 720 128 10 >> $O/ab.txt 2>&1 &&
  RSRT_COOP_DEFER=$d timeout -k 10 200 python tools/scene_time.py grid4 1280 720 32 10 >> $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
done
grep -v "^$" $O/ab.txt | awk 'NR%3==0'
GRID=$(python -c "import sys; sys.path.insert(0,'tools'); import make_big_scene; print(make_big_scene.make(4))")
for d in 40 1; do
RSRT_COOP_DEFER=$d timeout -k 10 200 python tools/simd_efficiency.py 4 32 suzanne 1280 720 10 >> $O/instr.txt 2>&1 &&
RSRT_COOP_DEFER=$d timeout -k 10 200 python tools/simd_efficiency.py 4 8 $GRID 1280 720 10 >> $O/instr.txt 2>&1 || { tail $O/instr.txt; exit 1; }
done
grep -v "descend\|leaf loop\|per TRACE inv\|outer trips\|SHADE: NEE" $O/instr.txt
