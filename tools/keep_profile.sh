#!/bin/bash
# bash tools/keep_profile.sh <tag>   (here, after tools/profile.sh <tag> ran on the GPU box): copies what is judged from
# gpurun_out/prof_<tag>/ into profiles/<tag>_*
R=$(cd "$(dirname "$0")/.." && pwd); T=$1; S=$R/gpurun_out/prof_$T
cp $S/bench.json $R/profiles/${T}_bench.json
cp $S/bench_under_rocprof.json $R/profiles/${T}_bench_under_rocprof.json
cp $S/summary.txt $R/profiles/${T}_rocprofv3_summary.txt
cp $(ls $S/trace/*/*_kernel_stats.csv | head -1) $R/profiles/${T}_kernel_stats.csv
case "$T" in *_house) cp $S/pmc_house_1080p_8b.json $R/profiles/pmc_house_1080p_8b.json;; esac
ls -la $R/profiles/${T}_*
