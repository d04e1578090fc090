#!/bin/bash
# bash tools/keep_profile.sh <tag>   (here, after tools/profile.sh <tag> ran on the GPU box): copies what is judged from
# gpurun_out/prof_<tag>/ into profiles/<tag>_*.  The kernel-stats CSV is the NEWEST one under the trace directory (profile.sh empties that
# directory before every run, so there is one) and gets a first line `# rsrt_build_id <id>` — the library the bench line of the same run
# reports — so that tests/test_bench_contract.py can hold it against the summary and against the committed sources.
R=$(cd "$(dirname "$0")/.." && pwd); T=$1; S=$R/gpurun_out/prof_$T
cp $S/bench.json $R/profiles/${T}_bench.json
cp $S/bench_under_rocprof.json $R/profiles/${T}_bench_under_rocprof.json
cp $S/summary.txt $R/profiles/${T}_rocprofv3_summary.txt
CSV=$(ls -t $S/trace/*/*_kernel_stats.csv | head -1)
N=$(ls $S/trace/*/*_kernel_stats.csv | wc -l)
[ "$N" = "1" ] || echo "keep_profile.sh: WARNING: $N kernel_stats.csv files under $S/trace (expected 1): taking the newest"
ID=$(python3 -c "import json,sys; print(json.loads([l for l in open('$S/bench_under_rocprof.json') if l.startswith('{')][-1])['roofline']['build_id'])")
{ echo "# rsrt_build_id $ID"; cat "$CSV"; } > $R/profiles/${T}_kernel_stats.csv
case "$T" in *_house) cp $S/pmc_house_1080p_8b.json $R/profiles/pmc_house_1080p_8b.json;; esac
ls -la $R/profiles/${T}_*
