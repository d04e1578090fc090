#!/bin/bash
# bash tools/pmc_set.sh "<counter list>" "<kernel variants>" [spp] — raw PMC sums for rt_render kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; CNT=$1; VARS=${2:-"3"}; SPP=${3:-64}
OUT=$R/gpurun_out/pmcset; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
for v in $VARS; do
  export RSRT_KERNEL=$v
  rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/v$v" -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp $SPP > "$OUT/v$v.json" 2> "$OUT/v$v.err" || { echo "variant $v failed"; tail -3 "$OUT/v$v.err"; }
  python3 - "$OUT/v$v" $v <<'PY'
import csv, glob, os, sys
a = {}
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rt_render' in r['Kernel_Name']:
            a[r['Counter_Name']] = a.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
print('variant', sys.argv[2], ' '.join('%s=%.4g' % kv for kv in sorted(a.items())))
PY
done
rm -rf "$OUT"
