#!/bin/bash
# round 4's closing GPU call: the whole GPU suite, smoke, the N = 2 rehearsal of bench.py, the ledger's ablations, the committed profiles
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.txt 2>&1; tail -3 gpurun_out/r04_gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/r04_multi_rehearsal.sh 2>&1 | grep "pre-flight\|frame_equals\|rc="
if ls rsoderh-raytracing_amd/librsrt_exp_*.so > /dev/null 2>&1; then bash tools/r04_ledger.sh 2>&1 | grep "ms/frame"; fi
bash tools/r04_profiles.sh > gpurun_out/r04_profiles_log.txt 2>&1; tail -3 gpurun_out/r04_profiles_log.txt
