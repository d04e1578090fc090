#!/bin/bash
# bash tools/walk_ab.sh ["knob configs"]   (GPU box) — the general-BVH scenes (tools/bvh_knobs.py: suzanne grid 16 spp, suzanne 64 spp) timed with the
# product library and with every experiment build librsrt_exp_*.so lying in the package directory, twice round the list.
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-RSRT_TRAVERSAL=4}
for pass in 1 2; do
  for lib in $R/rsoderh-raytracing_amd/librsrt.so $R/rsoderh-raytracing_amd/librsrt_exp_*.so; do
    [ -f "$lib" ] || continue
    RSRT_LIB=$lib python3 $R/tools/bvh_knobs.py "$CFG" | awk -v p=$pass '/^library/ {id=$2} / round 1 / {print "pass " p, id, $0}'
  done
done
