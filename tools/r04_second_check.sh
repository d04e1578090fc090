#!/bin/bash
# after the leaf trips went to record granularity: counters of the two general-BVH configs, the differential fuzz (small scenes forced through the
# cooperative walk as well), the big scenes
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_second_check; mkdir -p $O
bash tools/r04_pmc.sh > $O/pmc.txt 2>&1 || { tail $O/pmc.txt; exit 1; }
grep "ms/frame" $O/pmc.txt
timeout -k 10 400 python tools/fuzz_parity.py 600 4000 > $O/fuzz.txt 2>&1 || { tail $O/fuzz.txt; exit 1; }
tail -2 $O/fuzz.txt
RSRT_FLAT=0 timeout -k 10 400 python tools/fuzz_parity.py 400 7000 > $O/fuzz_noflat.txt 2>&1 || { tail $O/fuzz_noflat.txt; exit 1; }
tail -2 $O/fuzz_noflat.txt
timeout -k 10 500 python tools/big_scene_check.py 8 16 > $O/big.txt 2>&1 || { tail $O/big.txt; exit 1; }
tail -12 $O/big.txt
