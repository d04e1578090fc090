#!/bin/bash
# Runs on the GPU box:  bash tools/profile.sh <tag> [bench args, e.g. --scene suzanne --width 1280 --height 720 --spp 128 --bounces 10]
# 1) rocprofv3 --kernel-trace --stats of the bench command (no counters in this run);
# 2) the same bench command on its own: it collects the PMC counters itself, in separate rocprofv3 --pmc passes of a
#    child process (bench.py, collect_pmc), and prints the JSON line with the roofline object;
# 3) a text summary of both under gpurun_out/prof_<tag>/ (copied to profiles/ by hand).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}; shift || true
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
rm -rf "$OUT/trace"   # (a trace directory that accumulates runs was how a stale kernel_stats.csv got committed twice: one run, one set of files)
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $R/bench.py --no-pmc --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1 "$@" \
  > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" || { tail -5 "$OUT/bench_under_rocprof.err"; exit 1; }
timeout -k 10 400 python3 $R/bench.py --steps 5 --warmup 2 --write-profile "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cp $R/profiles/pmc_house_1080p_8b.json "$OUT/" 2>/dev/null
python3 $R/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +2M -delete
