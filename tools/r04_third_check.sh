#!/bin/bash
# closing assurance run on the final kernels: a longer differential fuzz (product choice, and every scene forced through the cooperative walk), the
# 991 k-triangle scene
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_third_check; mkdir -p $O
timeout -k 10 900 python tools/fuzz_parity.py 2400 20000 > $O/fuzz.txt 2>&1 || { tail $O/fuzz.txt; exit 1; }
tail -1 $O/fuzz.txt
RSRT_FLAT=0 timeout -k 10 900 python tools/fuzz_parity.py 2400 30000 > $O/fuzz_noflat.txt 2>&1 || { tail $O/fuzz_noflat.txt; exit 1; }
tail -1 $O/fuzz_noflat.txt
timeout -k 10 900 python tools/big_scene_check.py 32 > $O/big32.txt 2>&1 || { tail $O/big32.txt; exit 1; }
tail -7 $O/big32.txt
