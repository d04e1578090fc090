#!/bin/bash
# round 4: rocprofv3 summaries + bench lines (house with its live PMC counters and extra configs; suzanne; the 15 k-triangle grid), stage shares
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_profiles
mkdir -p $O
cd $R
timeout -k 10 200 python tools/simd_efficiency.py 4 64 > $O/simd_house.txt 2>&1; tail -12 $O/simd_house.txt
cp gpurun_out/stage_shares_house.json $O/house_stage_shares.json
bash tools/profile.sh r04_house 2>&1 | tail -40
GRID=$(python -c "import sys; sys.path.insert(0,'tools'); import make_big_scene; print(make_big_scene.make(4))")
bash tools/profile.sh r04_suzanne --scene suzanne --width 1280 --height 720 --spp 128 --bounces 10 2>&1 | grep -A3 '"utilisation"\|^{"metric' | head -20
bash tools/profile.sh r04_grid --scene $GRID --width 1280 --height 720 --spp 32 --bounces 10 2>&1 | grep -A3 '"utilisation"\|^{"metric' | head -20
