#!/bin/bash
# PMC counters of the two general-BVH configs, cooperative walk (6) against per-lane wide walk (4)
GRID=$(python -c "import sys; sys.path.insert(0,'tools'); import make_big_scene; print(make_big_scene.make(4))")
for t in 6 4; do
  RSRT_TRAVERSAL=$t timeout -k 10 280 bash tools/pmc_scene.sh suz_t$t suzanne 1280 720 128 10 "" || exit 1
  RSRT_TRAVERSAL=$t timeout -k 10 280 bash tools/pmc_scene.sh grid_t$t $GRID 1280 720 32 10 "" || exit 1
done
