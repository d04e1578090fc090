#!/bin/bash
# Diagnosis: bench.py on one scene with extra PMC passes.  bash tools/pmc_scene.sh <tag> <scene> <w> <h> <spp> <bounces> "<extra passes>"
# NO TA_* / TD_* counters in <extra passes>: a rocprofv3 --pmc run with them hung on this pool (round 2, a 15-minute call lost); refused here.
R=${GRAFT_REPO_ROOT:-/root/repo}
case "$7" in *TA_*|*TD_*) echo "pmc_scene.sh: TA_* / TD_* counters hang rocprofv3 on this pool: not run"; exit 2;; esac
TAG=$1; SCENE=$2; W=$3; H=$4; SPP=$5; MB=$6; EXTRA=$7
mkdir -p $R/gpurun_out
RSRT_PMC_EXTRA="$EXTRA" python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --scene $SCENE --width $W --height $H --spp $SPP --bounces $MB \
  > $R/gpurun_out/pmc_scene_$TAG.json 2> $R/gpurun_out/pmc_scene_$TAG.err || tail -5 $R/gpurun_out/pmc_scene_$TAG.err
python3 - <<PY
import json
b = json.load(open("$R/gpurun_out/pmc_scene_$TAG.json"))
ro = b["roofline"]; c = ro.get("counters") or {}
cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
print("$TAG: %.2f ms/frame, %.0f Mrays/s, frac %s, kernel %s, cycles %.3e" % (b["ms_per_frame"], b["value"], ro.get("frac"), ro.get("kernel"), cyc))
for k in sorted(c):
    print("   %-40s %16.0f   per CU-cycle %.4f" % (k, c[k], c[k] / (cyc * 256) if cyc else 0))
PY
