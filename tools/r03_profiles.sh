#!/bin/bash
# round 3: profiles (house, suzanne, grid) + stage shares + interactive with the queue hint + config table
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_call11
mkdir -p $O
cd $R
timeout -k 10 200 python tools/simd_efficiency.py 4 64 > $O/simd_house.txt 2>&1; tail -12 $O/simd_house.txt
mkdir -p profiles; cp gpurun_out/stage_shares_house.json profiles/r03_house_stage_shares.json
bash tools/profile.sh r03_house 2>&1 | tail -60
python -c "import sys; sys.path.insert(0,'tools'); import make_big_scene; make_big_scene.make(4)"
bash tools/profile.sh r03_suzanne --scene suzanne --width 1280 --height 720 --spp 128 --bounces 10 2>&1 | grep -A3 '"utilisation"\|^{"metric' | head -20
bash tools/profile.sh r03_grid --scene /tmp/rsrt_scenes/suzanne_grid_4.toml --width 1280 --height 720 --spp 32 --bounces 10 2>&1 | grep -A3 '"utilisation"\|^{"metric' | head -20
timeout -k 10 200 python tools/interactive_ab.py 1 16 > $O/interactive_ab.txt 2>&1; cat $O/interactive_ab.txt
timeout -k 10 400 python tools/config_table.py > $O/config_table.txt 2>&1; cat $O/config_table.txt; cp $R/gpurun_out/config_table.json $O/ 2>/dev/null
