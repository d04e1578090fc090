#!/bin/bash
# bash tools/trace_rate.sh [million rays]   (GPU box) — see tools/trace_rate.py
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_rate; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $R/tools/trace_rate.py "${1:-4}" 5 > "$OUT/run.txt" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
cat "$OUT/run.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
ln = [l for l in open(out + '/run.txt') if l.startswith('RAYS_PER_LAUNCH')][0].split(); n, rep = int(ln[1]), int(ln[3])
rows = [r for f in glob.glob(out + '/trace/*/*_kernel_trace.csv') for r in csv.DictReader(open(f)) if 'rt_cast_rays_kernel' in r['Kernel_Name']]
big = [r for r in rows if int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r['Grid_Size']) >= n]
for r in big[-5:]:
    ns = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print('%s  VGPR %s  LDS %s  %.3f ms  %.1f Grays/s (x%d traversals per ray)' % (r['Kernel_Name'][:60], r.get('VGPR_Count', '?'), r.get('LDS_Block_Size', '?'), ns / 1e6, n * rep / ns, rep))
PY
find "$OUT" -name "*.csv" -size +2M -delete
