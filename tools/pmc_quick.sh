#!/bin/bash
# bash tools/pmc_quick.sh <tag> "<variants>" [spp]   — SQ utilisation counters per kernel variant (1 frame each)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-q}; VARS=${2:-"0 1 2"}; SPP=${3:-64}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
for v in $VARS; do
  export RSRT_KERNEL=$v
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/v${v}a" -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp $SPP > "$OUT/v${v}a.json" 2> "$OUT/v${v}a.err" || echo "fail a $v"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d "$OUT/v${v}b" -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp $SPP > "$OUT/v${v}b.json" 2> "$OUT/v${v}b.err" || echo "fail b $v"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/v${v}c" -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp $SPP > "$OUT/v${v}c.json" 2> "$OUT/v${v}c.err" || echo "fail c $v"
done
python3 - "$OUT" "$VARS" <<'PY'
import csv, glob, os, sys
out, vars_ = sys.argv[1], sys.argv[2].split()
for v in vars_:
    a = {}
    for f in glob.glob(os.path.join(out, 'v%s?' % v, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rt_render' in r['Kernel_Name']:
                a[r['Counter_Name']] = a.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    if not a: print('variant', v, 'no data'); continue
    g = a.get('GRBM_GUI_ACTIVE', 0) / 8
    print('variant %s: kernel cycles/XCD %.3e  VALUBusy %.1f%%  lanes/VALU-inst %.1f  INSTS_VALU %.3e  SALU %.3e  LDS insts %.3e  bank-conflict %.1f%% of LDS active  WAIT_ANY %.0f%%  WAIT_INST %.0f%%  ACTIVE_ANY %.0f%% of wave-cycles  waves %d  VMEM_RD %.3e  LDS-array util %.1f%% (idx_active %.3e, conflict %.3e)  ACTIVE_INST_LDS %.3e' % (
        v, g, 100 * a.get('SQ_ACTIVE_INST_VALU', 0) / 256 / max(g, 1), a.get('SQ_THREAD_CYCLES_VALU', 0) / max(a.get('SQ_ACTIVE_INST_VALU', 1), 1),
        a.get('SQ_INSTS_VALU', 0), a.get('SQ_INSTS_SALU', 0), a.get('SQ_INSTS_LDS', 0), 100 * a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1),
        100 * a.get('SQ_WAIT_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1), 100 * a.get('SQ_WAIT_INST_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1),
        100 * a.get('SQ_ACTIVE_INST_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1), a.get('SQ_WAVES', 0), a.get('SQ_INSTS_VMEM_RD', 0), 100 * a.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / max(g, 1), a.get('SQ_LDS_IDX_ACTIVE', 0), a.get('SQ_LDS_BANK_CONFLICT', 0), a.get('SQ_ACTIVE_INST_LDS', 0)))
PY
find "$OUT" -name "*.csv" -delete
