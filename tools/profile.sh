#!/bin/bash
# Runs on the GPU box:  bash tools/profile.sh <tag> [extra bench args]
# 1) rocprofv3 --kernel-trace --stats of the default bench command, 2) PMC passes in separate runs
# (counters only with --kernel-trace, as the pool requires), 3) a text summary under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}; shift || true
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -- $BENCH --steps 1 --warmup 0 > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err" || echo "pmc set $i failed: $set"
done
python3 $R/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
python3 $R/tools/summarize_traffic.py "$OUT" "$OUT/hbm_traffic_house_1080p_8b.json" > /dev/null 2>&1
cat "$OUT/summary.txt"
# keep the merge small
find "$OUT" -name "*.csv" -size +2M -delete
