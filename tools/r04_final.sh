#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.txt 2>&1; tail -3 gpurun_out/r04_gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/r04_ledger.sh 2>&1 | tail -12
bash tools/r04_profiles.sh > gpurun_out/r04_profiles_log.txt 2>&1; tail -3 gpurun_out/r04_profiles_log.txt
